"""Builds libcrbm_hip.so (gfx950) in-tree with hipcc.

    python -m crbm_amd.csrc.build        # or: python crbm_amd/csrc/build.py

The host API and the model-independent kernels are compiled ahead of time into
crbm_amd/csrc/libcrbm_hip.so; the model-dependent kernels (crbm_kernels.h) are
specialised per model with hiprtc when a handle is created and cached under
crbm_amd/csrc/jit_cache/ (precompile() fills the cache without a GPU).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcrbm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
HEADERS = ["crbm_kernels.h", "crbm_kernels_generic.h", "crbm_layout.h", "crbm_jit.h", os.path.join("..", "..", "include", "crbm_amd.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(job):
    src, obj, extra = job
    deps = [os.path.join(HERE, src), os.path.abspath(__file__)] + [os.path.join(HERE, h) for h in HEADERS]
    if not _newer(obj, deps):
        return obj, False
    cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(verbose=True, jobs=None):
    os.makedirs(OBJ, exist_ok=True)
    work = [("crbm_api.hip", os.path.join(OBJ, "crbm_api.o"), [])]
    jobs = jobs or min(8, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, work))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lhiprtc", "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


def prune_cache(verbose=True):
    """Code objects are keyed by the full kernel source text: every file older than the newest kernel source was built
    from another revision (or another option set) and can never be hit again -- delete those, so that the in-tree
    cache that travels to the GPU box holds one revision."""
    cache = os.path.join(HERE, "jit_cache")
    if not os.path.isdir(cache):
        return 0
    newest = max(os.path.getmtime(os.path.join(HERE, h)) for h in ("crbm_kernels.h", "crbm_layout.h", "crbm_jit.h"))
    gone = 0
    for f in os.listdir(cache):
        path = os.path.join(cache, f)
        if f.endswith((".hsaco", ".tmp")) and os.path.getmtime(path) < newest:
            os.unlink(path)
            gone += 1
    if verbose and gone:
        print("jit cache: removed %d stale code objects" % gone)
    return gone


def precompile(configs, verbose=True):
    """JIT-compile the kernels of the given models into the on-disk cache.
    configs: iterable of dicts with num_motifs, motif_length, doublestranded,
    batchsize, fantasy_hidden_len.  Needs no GPU."""
    import ctypes
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from crbm_amd import _lib
    lib = _lib.load()
    for c in configs:
        cfg = _lib.CrbmConfig(num_motifs=c["num_motifs"], motif_length=c["motif_length"], input_dims=4,
                              doublestranded=int(c.get("doublestranded", 0)), batchsize=c.get("batchsize", 20),
                              cd_k=1, pooling=c.get("pooling", 1), fantasy_hidden_len=c.get("fantasy_hidden_len", 200),
                              learning_rate=0.1, momentum=0.9, rho=0.01, lambda_rate=0.1, seed=0, device=0, reserved=0)
        rc = lib.crbm_precompile(ctypes.byref(cfg))
        if rc != 0:
            raise RuntimeError("precompile failed for %s: %s" % (c, lib.crbm_last_error(None).decode()))
        if verbose:
            print("jit cache ready:", c)


if __name__ == "__main__":
    build()
