"""`CRBM` -- the reference's Python surface (secomo/convRBM.py:25-726) over the
HIP C-ABI (include/crbm_amd.h).

Same constructor signature, same validation and messages, same methods
(`fit`, `freeEnergy`, `motifHitProbs`, `getPFMs`, `saveModel`, `loadModel`) and
the `motifs` / `bias` / `c` objects with `get_value()` / `set_value()`; plus
`trainModel` (alias of `fit`, the name BASELINE.json's north_star uses).
Every tensor operation the reference hands to Theano runs in the HIP library;
this file only holds host logic (argument checks, batching, printing, pickles).
"""
import ctypes
import os
import time
import warnings

import joblib
import numpy as np
import scipy.stats

from . import _lib
from ._lib import CrbmConfig, as_f32, fptr


class _DeviceShared(object):
    """Stand-in for a `theano.shared` variable (convRBM.py:133,149,152):
    `get_value()` / `set_value()` on a parameter that lives on the GPU."""

    def __init__(self, model, name, shape):
        self._model = model
        self.name = name
        self._shape = shape

    def get_value(self, borrow=False):
        return self._model._get_param(self.name).copy()

    def set_value(self, value, borrow=False):
        value = as_f32(value)
        if value.shape != self._shape:
            raise ValueError("%s: expected shape %s, got %s" % (self.name, self._shape, value.shape))
        self._model._set_param(self.name, value)

    def __repr__(self):
        return self.name


class CRBM(object):
    """Convolutional RBM for DNA motifs; API of secomo.CRBM (convRBM.py:25-67).

    Parameters follow convRBM.py:68-71.  Keyword-only extras (not in the
    reference): ``fantasy_hidden_len`` (the reference hard-codes the hidden
    length of the persistent chains to 200, convRBM.py:168), ``seed``
    (reference: wall clock, convRBM.py:155) and ``device``.
    """

    def __init__(self, num_motifs, motif_length, epochs=100, input_dims=4,
                 doublestranded=True, batchsize=20, learning_rate=0.1,
                 momentum=0.95, pooling=1, cd_k=5,
                 rho=0.01, lambda_rate=0.1, **extra):
        # sanity checks: convRBM.py:72-108 (same order, same messages)
        if num_motifs <= 0:
            raise Exception("Number of motifs must be positive.")
        if motif_length <= 0:
            raise Exception("Motif length must be positive.")
        if epochs < 0:
            raise Exception("Epochs must be non-negative.")
        if input_dims <= 0:
            raise Exception("input_dims must be positive.")
        elif input_dims != 4:
            warnings.warn("input_dims != 4 was not comprehensively "
                          "tested yet. Be careful when interpreting the results.",
                          UserWarning)
        if batchsize <= 0:
            raise Exception("batchsize must be positive.")
        if learning_rate <= 0.0:
            raise Exception("learning_rate must be positive.")
        if not (momentum >= 0.0 and momentum < 1.):
            raise Exception("momentum must be between zero and one.")
        if pooling <= 0:
            raise Exception("pooling must be positive.")
        if cd_k <= 0:
            raise Exception("cd_k must be positive.")
        if not (rho >= 0.0 and rho < 1.):
            raise Exception("rho must be between zero and one.")
        if lambda_rate < 0.:
            raise Exception("lambda_rate must be non-negative.")

        fantasy_hidden_len = int(extra.pop("fantasy_hidden_len", 200))
        seed = extra.pop("seed", None)
        device = extra.pop("device", None)
        if extra:
            raise TypeError("unexpected keyword arguments: %s" % sorted(extra))
        # The reference takes any positive num_motifs and motif_length (convRBM.py:72-108); so does the library: models
        # beyond its specialised kernels (256 motifs, 64 letters, tables + one chain in the LDS) run on generic ones.
        # What remains is a capacity limit (README "Limits"): refuse at construction, not in the middle of fit().
        if num_motifs > 65536:
            raise Exception("num_motifs > 65536 is not supported.")
        if motif_length > 512:
            raise Exception("motif_length > 512 is not supported.")
        # any alphabet (the reference warns and runs: convRBM.py:84-87); other than DNA's four letters on the generic kernels
        if input_dims > 64:
            raise Exception("input_dims > 64 is not supported.")
        if input_dims * motif_length > 2048:
            raise Exception("input_dims * motif_length > 2048 is not supported.")

        # convRBM.py:111-123
        self.num_motifs = num_motifs
        self.motif_length = motif_length
        self.input_dims = input_dims
        self.doublestranded = doublestranded
        self.batchsize = batchsize
        self.learning_rate = learning_rate
        self.momentum = momentum
        self.rho = rho
        self.lambda_rate = lambda_rate
        self.pooling = pooling
        self.cd_k = cd_k
        self.epochs = epochs
        self.spmethod = 'entropy'
        self.fantasy_hidden_len = fantasy_hidden_len
        self.seed = int(time.time()) if seed is None else int(seed)      # :155
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device

        # convRBM.py:127-131: N(0,1) filters from NumPy's global RNG
        W = np.random.randn(num_motifs, 1, input_dims, motif_length).astype(np.float32)
        # convRBM.py:136-140
        if not rho:
            rho = 1. / (self.num_motifs * self.motif_length)
            if self.doublestranded:
                rho = rho / 2.
            self.rho = rho
        # convRBM.py:143-149 (floatX=float32; NumPy 2 would promote, so cast)
        b = (np.zeros((1, num_motifs)) +
             scipy.stats.norm.ppf(self.rho, 0, np.sqrt(motif_length))).astype(np.float32)
        c = np.zeros((1, input_dims), dtype=np.float32)                  # :151
        self._host = {"motifs": W, "bias": b, "c": c}
        self._handle = None
        self._control = None          # crbm_amd.dist.ControlPlane of a data-parallel job
        self._allreduce = None        # "rccl" or "ipc" once crbm_amd.dist.attach has made this model a rank
        self._pending_state = None    # velocities / chains / counters to install when the handle is created
        self.motifs = _DeviceShared(self, "motifs", W.shape)
        self.bias = _DeviceShared(self, "bias", b.shape)
        self.c = _DeviceShared(self, "c", c.shape)
        # data-parallel state (set by crbm_amd.dist.attach)
        self.world_size = 1
        self.rank = 0

    # ------------------------------------------------------------------ device
    def _h(self):
        """The device handle; created on first use.  Replaces the Theano
        compile step (convRBM.py:175, :453-515).  No CPU fallback."""
        if self._handle is not None:
            return self._handle
        lib = _lib.load()
        if self.batchsize % self.world_size != 0:
            raise Exception("batchsize must be divisible by the number of GPUs")
        cfg = CrbmConfig(
            num_motifs=self.num_motifs, motif_length=self.motif_length, input_dims=self.input_dims,
            doublestranded=1 if self.doublestranded else 0,
            batchsize=self.batchsize // self.world_size, cd_k=self.cd_k, pooling=self.pooling,
            fantasy_hidden_len=self.fantasy_hidden_len, learning_rate=self.learning_rate,
            momentum=self.momentum, rho=self.rho, lambda_rate=self.lambda_rate,
            seed=self.seed & 0xFFFFFFFFFFFFFFFF, device=self.device, reserved=0)
        handle = ctypes.c_void_p()
        rc = lib.crbm_create(ctypes.byref(cfg), ctypes.byref(handle))
        if rc != 0:
            raise Exception("crbm_create failed (%d): %s" % (rc, lib.crbm_last_error(None).decode()))
        self._handle = handle
        self._lib = lib
        self._call("crbm_set_params", fptr(self._host["motifs"]), fptr(self._host["bias"]),
                                        fptr(self._host["c"]))
        self._call("crbm_set_shard", self.rank * (self.batchsize // self.world_size))
        if self._pending_state is not None:
            st, self._pending_state = self._pending_state, None
            self._install_state(st)
        return handle

    def _call(self, name, *args):
        h = self._h()
        self._check(getattr(self._lib, name)(h, *args))

    def _check(self, rc):
        if rc != 0:
            raise Exception("HIP CRBM call failed (%d): %s"
                            % (rc, self._lib.crbm_last_error(self._handle).decode()))

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                self._lib.crbm_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _get_param(self, name):
        if self._handle is None:
            return self._host[name]
        W = np.empty((self.num_motifs, 1, self.input_dims, self.motif_length), dtype=np.float32)
        b = np.empty((1, self.num_motifs), dtype=np.float32)
        c = np.empty((1, self.input_dims), dtype=np.float32)
        self._check(self._lib.crbm_get_params(self._handle, fptr(W), fptr(b), fptr(c)))
        return {"motifs": W, "bias": b, "c": c}[name]

    def _set_param(self, name, value):
        if self._handle is None:
            self._host[name] = value
            return
        cur = {n: self._get_param(n) for n in ("motifs", "bias", "c")}
        cur[name] = value
        self._check(self._lib.crbm_set_params(self._handle, fptr(cur["motifs"]), fptr(cur["bias"]),
                                              fptr(cur["c"])))

    def _data(self, data):
        data = as_f32(data)
        if data.ndim != 4 or data.shape[1] != 1 or data.shape[2] != self.input_dims:
            raise Exception("expected a one-hot array of shape (n,1,%d,L), got %s" % (self.input_dims, data.shape))
        return data

    @staticmethod
    def _is_codes(data):
        """True for the packed input form: a 2-D uint8 array (n, L) of letter codes 0..input_dims-1
        (DNA: crbm_amd.sequences.seqsToCodes / fastaToCodes)."""
        return isinstance(data, np.ndarray) and data.ndim == 2 and data.dtype == np.uint8

    def _input(self, data):
        """-> (suffix, leading ctypes args, n, L) for the `crbm_*` / `crbm_*_codes` entry points."""
        if self._is_codes(data):
            codes = np.ascontiguousarray(data)
            return "_codes", (codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), codes.shape[0], codes.shape[1]), \
                codes.shape[0], codes.shape[1], codes
        data = self._data(data)
        return "", (fptr(data), data.shape[0], data.shape[3]), data.shape[0], data.shape[3], data

    # ------------------------------------------------------------- persistence
    def saveModel(self, filename):
        """convRBM.py:177-204 -- same pickle tuple."""
        numpyParams = (self.motifs.get_value(), self.bias.get_value(), self.c.get_value())
        hyperparams = (self.num_motifs, self.motif_length, self.input_dims, self.doublestranded,
                       self.batchsize, self.learning_rate, self.momentum, self.rho, self.lambda_rate,
                       self.pooling, self.cd_k, self.epochs, self.spmethod)
        joblib.dump((numpyParams, hyperparams), filename, protocol=2)

    @classmethod
    def loadModel(cls, filename):
        """convRBM.py:206-236 -- velocities and chains restart from zero."""
        numpyParams, hyperparams = joblib.load(filename)
        (num_motifs, motif_length, input_dims, doublestranded, batchsize, learning_rate,
         momentum, rho, lambda_rate, pooling, cd_k, epochs, spmethod) = hyperparams
        obj = cls(num_motifs, motif_length, epochs=epochs, input_dims=input_dims,
                  doublestranded=doublestranded, batchsize=batchsize, learning_rate=learning_rate,
                  momentum=momentum, pooling=pooling, cd_k=cd_k, rho=rho, lambda_rate=lambda_rate)
        motifs, bias, c = numpyParams
        obj.motifs.set_value(motifs)
        obj.bias.set_value(bias)
        obj.c.set_value(c)
        return obj

    def _full_state(self):
        """velocities, chains of ALL ranks (bit-packed) and sampler counters; collective in a
        data-parallel job (every rank calls it, every rank gets the global state)."""
        h = self._h()
        seed, gstep, estep = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32()
        self._check(self._lib.crbm_get_rng(h, ctypes.byref(seed), ctypes.byref(gstep), ctypes.byref(estep)))
        fh, fhp = self.get_fantasy()
        mine = (np.packbits(fh.astype(np.uint8), axis=None),
                None if fhp is None else np.packbits(fhp.astype(np.uint8), axis=None))
        parts = [mine] if self.world_size == 1 else self._control.gather(mine)
        shape = (self.batchsize, self.num_motifs, 1, self.fantasy_hidden_len)
        per = (self.batchsize // self.world_size) * self.num_motifs * self.fantasy_hidden_len

        def unpack(i):
            if parts[0][i] is None:
                return None
            rows = [np.unpackbits(p[i], count=per) for p in parts]
            return np.packbits(np.concatenate(rows).reshape(shape), axis=3)   # along Lf, one byte per 8 positions
        return {"velocities": self.get_velocities(), "fantasy_bits": (unpack(0), unpack(1)),
                "fantasy_hidden_len": self.fantasy_hidden_len, "world_size": self.world_size,
                "rng": (int(seed.value), int(gstep.value), int(estep.value))}

    def saveState(self, filename):
        """Full training state: the reference tuple of
        saveModel plus velocities, persistent chains and the sampler counters,
        so that training resumes exactly (the reference's loadModel restarts
        momentum and chains from zero, convRBM.py:226-235).  A joblib file whose
        first two entries are the reference's (numpyParams, hyperparams).  In a
        data-parallel job every rank calls it; the chains of all ranks are
        gathered and rank 0 writes the one file, which loads at any world size."""
        numpyParams = (self.motifs.get_value(), self.bias.get_value(), self.c.get_value())
        hyperparams = (self.num_motifs, self.motif_length, self.input_dims, self.doublestranded,
                       self.batchsize, self.learning_rate, self.momentum, self.rho, self.lambda_rate,
                       self.pooling, self.cd_k, self.epochs, self.spmethod)
        extra = self._full_state()
        if self.rank == 0:
            joblib.dump((numpyParams, hyperparams, extra), filename, protocol=2)
        if self._control is not None:
            self._control.barrier()

    @classmethod
    def loadState(cls, filename):
        """Counterpart of saveState.  The device state (velocities, chains, counters) is
        installed when the handle is created, so `crbm_amd.dist.attach()` can still be called
        on the returned model; each rank then takes its own chains out of the global state."""
        numpyParams, hyperparams, extra = joblib.load(filename)
        (num_motifs, motif_length, input_dims, doublestranded, batchsize, learning_rate,
         momentum, rho, lambda_rate, pooling, cd_k, epochs, spmethod) = hyperparams
        seed, gstep, estep = extra["rng"]
        obj = cls(num_motifs, motif_length, epochs=epochs, input_dims=input_dims,
                  doublestranded=doublestranded, batchsize=batchsize, learning_rate=learning_rate,
                  momentum=momentum, pooling=pooling, cd_k=cd_k, rho=rho, lambda_rate=lambda_rate,
                  fantasy_hidden_len=extra["fantasy_hidden_len"], seed=seed)
        motifs, bias, c = numpyParams
        obj.motifs.set_value(motifs)
        obj.bias.set_value(bias)
        obj.c.set_value(c)
        if "fantasy_bits" not in extra:      # files written before the chains were bit-packed
            fh, fhp = extra["fantasy"]
            extra = dict(extra, fantasy_bits=(np.packbits(fh.astype(np.uint8), axis=3),
                                              None if fhp is None else np.packbits(fhp.astype(np.uint8), axis=3)))
        obj._pending_state = {"velocities": extra["velocities"], "fantasy_bits": extra["fantasy_bits"],
                              "rng": (seed, gstep, estep)}
        return obj

    def _install_state(self, st):
        nb = self.batchsize // self.world_size
        lo = self.rank * nb

        def rows(bits):
            if bits is None:
                return None
            if bits.shape[0] != self.batchsize:
                raise Exception("saved state holds %d chains, the model has %d" % (bits.shape[0], self.batchsize))
            return np.unpackbits(bits[lo:lo + nb], axis=3, count=self.fantasy_hidden_len).astype(np.float32)
        self.set_velocities(*st["velocities"])
        fb, fbp = st["fantasy_bits"]
        self.set_fantasy(rows(fb), rows(fbp))
        seed, gstep, estep = st["rng"]
        self.set_rng(seed, gstep, estep)

    # ------------------------------------------- graph builders as plain calls
    def _hgv(self, data, flip, want, rng_step=0):
        data = self._data(data)
        n, L = data.shape[0], data.shape[3]
        shape = (n, self.num_motifs, 1, L - self.motif_length + 1)
        outs = [np.empty(shape, dtype=np.float32) if w else None for w in want]
        self._call("crbm_h_given_v", fptr(data), n, L, 1 if flip else 0,
                                             rng_step, fptr(outs[0]), fptr(outs[1]), fptr(outs[2]))
        return outs

    def _bottomUpActivity(self, data, flip_motif=False):
        """convRBM.py:238-243."""
        return self._hgv(data, flip_motif, (True, False, False))[0]

    def _bottomUpProbabilityOfData(self, data, flip_motif=False):
        """_bottomUpProbability(_bottomUpActivity(data)) (convRBM.py:245-257)."""
        return self._hgv(data, flip_motif, (False, True, False))[1]

    def _computeHgivenV(self, data, flip_motif=False, rng_step=0):
        """convRBM.py:269-275 -> [probability, sample]."""
        o = self._hgv(data, flip_motif, (False, True, True), rng_step)
        return [o[1], o[2]]

    def _vgh(self, h, hprime, want, rng_step=0):
        h = as_f32(h)
        hp = None if hprime is None else as_f32(hprime)
        if h.ndim != 4 or h.shape[1] != self.num_motifs or h.shape[2] != 1:
            raise Exception("expected hidden array of shape (n,K,1,Lh), got %s" % (h.shape,))
        if hp is not None and hp.shape != h.shape:
            raise Exception("h and hprime must have the same shape")
        n, Lh = h.shape[0], h.shape[3]
        shape = (n, 1, self.input_dims, Lh + self.motif_length - 1)
        outs = [np.empty(shape, dtype=np.float32) if w else None for w in want]
        self._call("crbm_v_given_h", fptr(h), fptr(hp), n, Lh, rng_step,
                                             fptr(outs[0]), fptr(outs[1]), fptr(outs[2]))
        return outs

    def _topDownActivity(self, h, hprime=None):
        """convRBM.py:277-292."""
        return self._vgh(h, hprime, (True, False, False))[0]

    def _topDownProbabilityOfHidden(self, h, hprime=None):
        """_topDownProbability(_topDownActivity(h, hprime)) (convRBM.py:294-299)."""
        return self._vgh(h, hprime, (False, True, False))[1]

    def _computeVgivenH(self, H_sample, H_sample_prime=None, rng_step=0):
        """convRBM.py:317-325 -> [probability, sample]."""
        o = self._vgh(H_sample, H_sample_prime, (False, True, True), rng_step)
        return [o[1], o[2]]

    # ------------------------------------------------------------ evaluation
    def _evaluateData(self, data):
        """convRBM.py:517-522 -> [mean free energy, mean of a sampled H]."""
        data = self._data(data)
        mfe, nmh = ctypes.c_float(), ctypes.c_float()
        self._call("crbm_eval_data", fptr(data), data.shape[0], data.shape[3],
                                             ctypes.byref(mfe), ctypes.byref(nmh))
        return [mfe.value, nmh.value]

    def _evaluateParams(self):
        """convRBM.py:528-533 -> [rms(W), IC, median IC]."""
        a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        self._call("crbm_eval_params", ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return [a.value, b.value, c.value]

    def _trainingFct(self, data):
        """convRBM.py:524-526: one PCD-k update on a mini-batch."""
        data = self._data(data)
        self._call("crbm_train_step", fptr(data), data.shape[0], data.shape[3])

    def motifHitProbs(self, data):
        """convRBM.py:535-547 -> (n,K,1,L-M+1).  `data` is the reference's one-hot
        array or, for data-set scale sweeps, (n,L) uint8 letter codes."""
        suffix, args, n, L, _keep = self._input(data)
        out = np.empty((n, self.num_motifs, 1, L - self.motif_length + 1), dtype=np.float32)
        self._call("crbm_hit_probs" + suffix, *(args + (fptr(out),)))
        return out

    def motifHitSummary(self, data, position_mean=True):
        """The reductions of motifHitProbs() the reference's analysis code takes
        (utils.py:113-116, :154, :242-244, :305), computed on the device without the
        dense (n,K,1,Lh) tensor: dict with 'max' (n,K) = P.max(axis=(2,3)),
        'mean' (n,K) = P.mean(axis=(2,3)), 'position_mean' (K,Lh) = P.mean(axis=(0,2))."""
        suffix, args, n, L, _keep = self._input(data)
        K, Lh = self.num_motifs, L - self.motif_length + 1
        hmax = np.empty((n, K), dtype=np.float32)
        hmean = np.empty((n, K), dtype=np.float32)
        pos = np.empty((K, Lh), dtype=np.float32) if position_mean else None
        self._call("crbm_hit_summary" + suffix, *(args + (fptr(hmax), fptr(hmean),
                                                           fptr(pos) if position_mean else None)))
        out = {"max": hmax, "mean": hmean}
        if position_mean:
            out["position_mean"] = pos
        return out

    def freeEnergy(self, data, permotif=False):
        """convRBM.py:549-568 -> (n,) or (n,K); one-hot or (n,L) uint8 codes."""
        suffix, args, n, L, _keep = self._input(data)
        out = np.empty((n, self.num_motifs) if permotif else (n,), dtype=np.float32)
        if suffix:
            self._call("crbm_free_energy_codes", *(args + ((None, fptr(out)) if permotif else (fptr(out), None))))
        else:
            self._call("crbm_free_energy_per_motif" if permotif else "crbm_free_energy", *(args + (fptr(out),)))
        return out

    def getPFMs(self):
        """Position frequency matrices of the filters: per motif a (4,M) float64 array whose
        columns are the softmax of the filter column over the four letters (what
        convRBM.py:640-655 returns; host NumPy there too)."""
        W = self.motifs.get_value().astype(np.float64)[:, 0]      # (K,4,M)
        e = np.exp(W)
        pfm = e / e.sum(axis=1, keepdims=True)
        return [pfm[k] for k in range(self.num_motifs)]

    # --------------------------------------------------------------- training
    def gibbsSteps(self, k=1):
        """Advance the persistent chains by k Gibbs steps (convRBM.py:397-408)
        with the parameters frozen."""
        self._call("crbm_gibbs_steps", int(k))

    def _upload(self, data, slot):
        """Make a data set resident in HBM (packed letters) in the given slot; returns (n, L).  Letter codes travel as they
        are (one byte per base); a float one-hot array is streamed to the device in slabs and encoded -- and checked to be
        exactly one-hot -- there: the PCIe copy of 16 bytes per base is an order of magnitude cheaper than any pass over the
        array on the host (config #1: 3 ms of NumPy for 0.2 ms of copy)."""
        self._call("crbm_dataset_select", slot)
        if self._is_codes(data):
            codes = np.ascontiguousarray(data)
            self._call("crbm_dataset_upload_codes", codes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                       codes.shape[0], codes.shape[1])
            return codes.shape
        data = self._data(data)
        self._call("crbm_dataset_upload", fptr(data), data.shape[0], data.shape[3])
        return (data.shape[0], data.shape[3])

    def _truncate(self, data):
        """convRBM.py:586-599: truncate so that (L-M+1) % pooling == 0."""
        L = data.shape[-1]
        nseq = int((L - self.motif_length + 1) / self.pooling) * self.pooling + self.motif_length - 1
        return data[..., :nseq]

    def fit(self, training_data, test_data=None):
        """convRBM.py:570-634: epochs x sequential, unshuffled mini-batches;
        one status line per epoch.  Both sets are uploaded once (packed 2-bit)
        and the loops only pass row bounds.  Besides the reference's one-hot
        arrays, (n,L) uint8 letter codes are accepted."""
        training_data = self._truncate(training_data)
        same = test_data is None
        test_data = training_data if same else self._truncate(test_data)

        sharded = self.world_size > 1
        evaluates = self.rank == 0          # data-parallel: rank 0 alone evaluates and prints
        if evaluates:
            print(("BatchSize: " + str(self.batchsize)))
            print("Start training the model...")
        starttime = time.time()
        self._h()
        ntrain = ntest = 0
        if self.epochs > 0:
            ntrain = training_data.shape[0]
            if sharded:
                # each rank uploads only the rows it owns of every mini-batch (1/world of the set)
                from . import dist
                # every rank knows every share: all of them refuse together (a lone raise on the empty
                # rank would leave the others waiting in the all-reduce for ever)
                empty = dist.empty_shards(ntrain, self.batchsize, self.world_size)
                if empty:
                    raise Exception("fewer training rows than ranks: rank(s) %s would own none of the %d rows"
                                    % (empty, ntrain))
                mine = dist.shard_rows(ntrain, self.batchsize, self.rank, self.world_size)
                self._upload(training_data[mine], 0)
                if evaluates:
                    ntest = self._upload(test_data, 1)[0]
            else:
                self._upload(training_data, 0)
                ntest = ntrain if same else self._upload(test_data, 1)[0]
        test_slot = 1 if (sharded or not same) else 0
        for epoch in range(self.epochs):
            self._call("crbm_dataset_select", 0)
            # the batch loop of convRBM.py:612-615 runs inside the library: the steps are
            # enqueued back to back, with one host synchronisation per epoch
            if sharded:
                # All ranks enter an epoch together: rank 0 alone converts and uploads the test set before its first
                # step and alone evaluates after every epoch (below), while the other ranks come straight back here.
                # Without this barrier their update launches would spend that time waiting ON THE DEVICE for rank 0's
                # sums -- a wait that the mapped-buffer all-reduce bounds (CRBM_IPC_TIMEOUT_MS) so that a dead peer
                # cannot hang a GPU.  Host skew belongs on the host: the control plane waits (CRBM_CONTROL_TIMEOUT).
                if self._control is not None:
                    self._control.barrier()
                # (a wait that ran out inside the epoch comes back as CRBM_ERR_IPC_TIMEOUT: no update was applied
                #  after it, the parameters are those of the last complete step)
                self._call("crbm_train_epoch_sharded", self.batchsize, ntrain, int(training_data.shape[-1]))
            else:
                self._call("crbm_train_epoch_resident", self.batchsize)
            if not evaluates:
                continue
            # convRBM.py:616-625 on the resident test set: the loop over its mini-batches (mean free energy and mean
            # sampled activity per batch, averaged over the batches) runs inside the library with one synchronisation
            # -- 50 round trips per epoch at the reference's defaults were 4 of the 6 ms an epoch of config #1 took
            self._call("crbm_dataset_select", test_slot)
            mfe, nmh = ctypes.c_double(), ctypes.c_double()
            self._call("crbm_eval_epoch_resident", self.batchsize, ctypes.byref(mfe), ctypes.byref(nmh))
            self._call("crbm_dataset_select", 0)
            [twn_, ic_, medic_] = self._evaluateParams()
            print(("Epoch {:d}: ".format(epoch) +
                   "FE={:1.3f} ".format(mfe.value) +
                   "NumH={:1.4f} ".format(nmh.value) +
                   "WNorm={:2.2f} ".format(float(twn_)) +
                   "IC={:1.3f} medIC={:1.3f}".format(float(ic_), float(medic_))))
        if evaluates:
            print(("Training finished after: {:5.2f} seconds!".format(time.time() - starttime)))

    # the name BASELINE.json's north_star uses for the same entry point
    trainModel = fit

    def _shard_rows(self, start, end):
        """Rows of mini-batch [start,end) this rank owns (contiguous, balanced)."""
        if self.world_size == 1:
            return start, end
        n = end - start
        lo = start + (n * self.rank) // self.world_size
        hi = start + (n * (self.rank + 1)) // self.world_size
        return lo, hi

    def _iterateBatchIndices(self, totalsize, nbatchsize):
        """convRBM.py:722-726."""
        return [[i, i + nbatchsize] if i + nbatchsize <= totalsize
                else [i, totalsize] for i in range(totalsize)[0::nbatchsize]]

    def __repr__(self):
        """Same text as the reference prints (convRBM.py:704-720): two headed sections, the
        hyper-parameters run together without separators."""
        fields = [("input dims: {:d}", self.input_dims), ("doublestranded: {}", self.doublestranded),
                  ("batchsize: {:d}", self.batchsize), ("learning rate: {:1.3f}", self.learning_rate),
                  ("momentum: {:1.3f}", self.momentum), ("rho: {:1.4f}", self.rho),
                  ("lambda: {:1.3f}", self.lambda_rate), ("pooling: {:d}", self.pooling),
                  ("cd_k: {:d}", self.cd_k), ("epochs: {:d}", self.epochs)]
        head = "Parameters:\n\nNumber of motifs: %s\nMotif length: %s\n\nHyper-parameters:\n\n" \
            % (self.num_motifs, self.motif_length)
        return head + "".join(fmt.format(val) for fmt, val in fields)

    # ------------------------------------------- state the reference never saves
    def get_velocities(self):
        vW = np.empty((self.num_motifs, 1, self.input_dims, self.motif_length), dtype=np.float32)
        vb = np.empty((1, self.num_motifs), dtype=np.float32)
        vc = np.empty((1, self.input_dims), dtype=np.float32)
        self._call("crbm_get_velocities", fptr(vW), fptr(vb), fptr(vc))
        return vW, vb, vc

    def set_velocities(self, vW, vb, vc):
        vW, vb, vc = as_f32(vW), as_f32(vb), as_f32(vc)
        want = ((self.num_motifs, 1, self.input_dims, self.motif_length), (1, self.num_motifs), (1, self.input_dims))
        for name, arr, shape in zip(("vW", "vb", "vc"), (vW, vb, vc), want):
            if arr.shape != shape:          # the C side reads exactly these many floats
                raise ValueError("%s: expected shape %s, got %s" % (name, shape, arr.shape))
        self._call("crbm_set_velocities", fptr(vW), fptr(vb), fptr(vc))

    def get_fantasy(self):
        """(fantasy_h, fantasy_h_prime or None), dense float32 (convRBM.py:168-173)."""
        h = self._h()
        nb = self.batchsize // self.world_size
        shape = (nb, self.num_motifs, 1, self.fantasy_hidden_len)
        a = np.empty(shape, dtype=np.float32)
        b = np.empty(shape, dtype=np.float32) if self.doublestranded else None
        self._call("crbm_get_fantasy", fptr(a), fptr(b))
        return a, b

    def set_fantasy(self, hid, hid_prime=None):
        """This rank's chains: (batchsize / world_size, K, 1, fantasy_hidden_len), exactly 0/1."""
        self._h()
        shape = (self.batchsize // self.world_size, self.num_motifs, 1, self.fantasy_hidden_len)
        hid = as_f32(hid)
        hid_prime = None if hid_prime is None else as_f32(hid_prime)
        for name, arr in (("hid", hid), ("hid_prime", hid_prime)):
            if arr is not None and arr.shape != shape:      # the C side reads exactly B*K*Lf floats
                raise ValueError("%s: expected shape %s, got %s" % (name, shape, arr.shape))
        if self.doublestranded and hid_prime is None:
            raise ValueError("a doublestranded model needs hid_prime")
        self._call("crbm_set_fantasy", fptr(hid), fptr(hid_prime))

    def get_fantasy_visible(self):
        h = self._h()
        nb = self.batchsize // self.world_size
        v = np.empty((nb, 1, self.input_dims, self.fantasy_hidden_len + self.motif_length - 1), dtype=np.float32)
        self._call("crbm_get_fantasy_visible", fptr(v))
        return v

    def get_rng(self):
        """(seed, gibbs_step, eval_step): the sampler's seed and its two step counters (counter word 3 of the chain draws and
        of the evaluation draws)."""
        seed, gstep, estep = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32()
        self._call("crbm_get_rng", ctypes.byref(seed), ctypes.byref(gstep), ctypes.byref(estep))
        return seed.value, gstep.value, estep.value

    def set_rng(self, seed=None, gibbs_step=0, eval_step=0):
        if seed is not None:
            self.seed = int(seed)
        self._call("crbm_set_rng", self.seed & 0xFFFFFFFFFFFFFFFF, gibbs_step, eval_step)
