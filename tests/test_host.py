"""CPU-only tests of the host side: the C-ABI library loads and exports every
symbol include/crbm_amd.h declares, the CRBM class validates like the
reference (tests/testcrbm.py:14-98), the product path never touches the oracle
and fails loudly without a GPU."""
import os
import re
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from crbm_amd import _lib
    header = open(os.path.join(ROOT, "include", "crbm_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(crbm_[a-z_]+)\s*\(", header))
    declared -= {"crbm_status", "crbm_config", "crbm_handle", "crbm_launch_info"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                      # binds every symbol or raises
    assert lib.crbm_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define CRBM_AMD_ABI_VERSION (\d+)", header).group(1))
    assert lib.crbm_device_count() >= 0


def test_crbm_parametervalues():
    """reference tests/testcrbm.py:14-56."""
    from crbm_amd import CRBM
    bad = [dict(num_motifs=0, motif_length=5), dict(num_motifs=1, motif_length=0),
           dict(num_motifs=1, motif_length=1, epochs=-1), dict(num_motifs=1, motif_length=1, input_dims=-1),
           dict(num_motifs=1, motif_length=1, batchsize=0), dict(num_motifs=1, motif_length=1, learning_rate=0.0),
           dict(num_motifs=1, motif_length=1, momentum=-1.), dict(num_motifs=1, motif_length=1, momentum=1.1),
           dict(num_motifs=1, motif_length=1, pooling=0), dict(num_motifs=1, motif_length=1, cd_k=0),
           dict(num_motifs=1, motif_length=1, rho=-1.), dict(num_motifs=1, motif_length=1, rho=1.1),
           dict(num_motifs=1, motif_length=1, lambda_rate=-.1)]
    for kw in bad:
        with pytest.raises(Exception):
            CRBM(**kw)
    with pytest.warns(UserWarning):
        CRBM(num_motifs=1, motif_length=1, input_dims=3)
    CRBM(num_motifs=1, motif_length=1, epochs=0)            # legal (convRBM.py:79)


def test_crbm_creation_and_reload(tmp_path):
    """reference tests/testcrbm.py:58-98 (host state only: no GPU call is made)."""
    from crbm_amd import CRBM
    model = CRBM(num_motifs=2, motif_length=5)
    W, b, c = model.motifs.get_value(), model.bias.get_value(), model.c.get_value()
    assert W.shape == (2, 1, 4, 5) and b.shape == (1, 2) and c.shape == (1, 4)
    assert W.dtype == np.float32 and b.dtype == np.float32
    fn = str(tmp_path / "model.pkl")
    model.saveModel(fn)
    model2 = CRBM.loadModel(fn)
    for attr in ("num_motifs", "motif_length", "epochs", "input_dims", "doublestranded", "batchsize",
                 "momentum", "pooling", "cd_k", "rho", "lambda_rate"):
        assert getattr(model, attr) == getattr(model2, attr)
    np.testing.assert_allclose(W, model2.motifs.get_value())
    np.testing.assert_allclose(b, model2.bias.get_value())
    np.testing.assert_allclose(c, model2.c.get_value())
    # reference pickle layout (convRBM.py:186-204)
    import joblib
    params, hyper = joblib.load(fn)
    assert len(params) == 3 and len(hyper) == 13 and hyper[-1] == "entropy"


def test_init_matches_reference_formulas():
    from crbm_amd import CRBM
    m = CRBM(10, 15)
    np.testing.assert_allclose(m.bias.get_value(), -9.0099, atol=1e-4)     # norm.ppf(0.01, 0, sqrt(15))
    assert CRBM(10, 15, rho=0.0).rho == 1.0 / 300                          # convRBM.py:136-140
    assert CRBM(10, 15, rho=0.0, doublestranded=False).rho == 1.0 / 150
    assert m._iterateBatchIndices(45, 20) == [[0, 20], [20, 40], [40, 45]]
    assert CRBM.trainModel is CRBM.fit
    assert "Number of motifs: 10" in repr(m)
    with pytest.raises(ValueError):
        m.motifs.set_value(np.zeros((3, 3)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "crbm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "oracle/" not in text.replace("oracle/crbm_oracle.py", ""), f


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_fails_loudly_without_gpu():
    from crbm_amd import CRBM
    m = CRBM(4, 5)
    with pytest.raises(Exception, match="no HIP device|no CPU"):
        m.motifHitProbs(np.zeros((1, 1, 4, 20), dtype=np.float32))
    with pytest.raises(Exception):
        m.fit(np.zeros((2, 1, 4, 20), dtype=np.float32))


def test_sequence_helpers():
    from crbm_amd import seqsToCodes, codesToOneHot, seqToOneHot
    seqs = ["ACGTac", "ttGGca"]
    codes = seqsToCodes(seqs)
    np.testing.assert_array_equal(codes, [[0, 1, 2, 3, 0, 1], [3, 3, 2, 2, 1, 0]])
    oh = seqToOneHot(seqs)
    assert oh.shape == (2, 1, 4, 6) and oh.dtype == np.float32          # sequences.py:114-117
    np.testing.assert_array_equal(oh.sum(axis=2), 1.0)
    assert oh[0, 0, 3, 3] == 1 and oh[1, 0, 0, 5] == 1                   # letter map sequences.py:9-17
    np.testing.assert_array_equal(codesToOneHot(codes), oh)
    with pytest.raises(Exception):
        seqsToCodes(["ACGN"])
    with pytest.raises(Exception):
        seqsToCodes(["ACG", "AC"])
    # another alphabet: the codes of an input_dims = 20 model (README "Limits")
    from crbm_amd import sequences as sq2
    p = seqsToCodes(["MKV", "acd"], sq2.PROTEIN)
    np.testing.assert_array_equal(p, [[10, 8, 17], [0, 1, 2]])
    assert codesToOneHot(p, 20).shape == (2, 1, 20, 3) and codesToOneHot(p, 20).sum() == 6
    with pytest.raises(Exception, match="may only contain"):
        seqsToCodes(["MKB"], sq2.PROTEIN)


# ---- sequences.py / utils.py counterparts ---------------------------
def test_fasta_reader_and_split(tmp_path, capsys):
    from crbm_amd import sequences as sq
    fa = tmp_path / "toy.fa"
    fa.write_text(">s1 first\nACGT\nacgt\n\n>s2\nACGTNCGT\n>s3 third one\nTTTTCCCC\n>s4\nGGGGAAAA\n")
    recs = sq.readSeqsFromFasta(str(fa))
    assert "skip sequence containing N" in capsys.readouterr().out          # sequences.py:47-50
    assert [r.id for r in recs] == ["s1", "s3", "s4"] and recs[0].seq == "ACGTacgt"
    assert recs[1].description == "s3 third one"
    codes = sq.fastaToCodes(str(fa))
    assert codes.dtype == np.uint8 and codes.shape == (3, 8)
    np.testing.assert_array_equal(codes[0], [0, 1, 2, 3, 0, 1, 2, 3])
    onehot = sq.seqToOneHot(recs)
    assert onehot.shape == (3, 1, 4, 8) and onehot.dtype == np.float32
    np.testing.assert_array_equal(onehot, sq.codesToOneHot(codes))
    np.testing.assert_array_equal(onehot.sum(axis=2), 1.0)
    sq.splitTrainingTest(str(fa), 0.34, randomize=False)                  # sequences.py:54-98
    tr = sq.readSeqsFromFasta(str(tmp_path / "toy_train.fa"))
    te = sq.readSeqsFromFasta(str(tmp_path / "toy_test.fa"))
    assert [r.id for r in te] == ["s1"] and [r.id for r in tr] == ["s3", "s4"]
    sq.splitTrainingTest(str(fa), 0.5, num_top_regions=2, randomize=True)
    assert len(sq.readSeqsFromFasta(str(tmp_path / "toy_train.fa"))) == 1
    with pytest.raises(Exception, match="same length"):
        sq.seqsToCodes(["ACGT", "ACG"])
    with pytest.raises(Exception, match="load_sample"):
        sq.load_sample(str(tmp_path / "missing.fa"))
    assert sq.load_sample(str(fa)).shape == (3, 1, 4, 8)


def test_save_motifs_formats(tmp_path):
    from crbm_amd.utils import saveMotifs, formatMotif

    class Fake(object):
        def getPFMs(self):
            return [np.array([[0.7, 0.1], [0.1, 0.2], [0.1, 0.3], [0.1, 0.4]]), np.full((4, 2), 0.25)]

    saveMotifs(Fake(), str(tmp_path / "out"), name="m")                    # utils.py:16-47
    txt = (tmp_path / "out" / "m0.pfm").read_text().splitlines()
    assert txt[0] == ">m1 m1"
    assert txt[1] == "A [  0.70   0.10]" and txt[4] == "T [  0.10   0.40]"
    assert (tmp_path / "out" / "m1.pfm").exists()
    assert formatMotif(np.full((4, 2), 0.25), "x", "pfm").splitlines()[0] == "  0.25   0.25"
    assert formatMotif(np.full((4, 2), 0.25), "x", "tab").splitlines()[2] == "G\t0.25\t0.25"
    with pytest.raises(ValueError):
        formatMotif(np.full((4, 2), 0.25), "x", "meme")


def test_codes_detection():
    from crbm_amd import CRBM
    assert CRBM._is_codes(np.zeros((3, 9), np.uint8))
    assert not CRBM._is_codes(np.zeros((3, 1, 4, 9), np.float32))
    assert not CRBM._is_codes(np.zeros((3, 9), np.float32))


def test_limits_are_refused_at_construction():
    """what the kernels cannot take must not wait for the middle of fit().  The reference accepts any
    positive K, M (convRBM.py:72-108); round 4 lifted the 256-motif / 64-letter / LDS bounds of the specialised
    kernels (generic kernels take over beyond them), so what is refused at construction is a capacity limit only."""
    from crbm_amd import CRBM
    with pytest.raises(Exception, match="num_motifs > 65536"):
        CRBM(65537, 5)
    with pytest.raises(Exception, match="motif_length > 512"):
        CRBM(4, 513)
    CRBM(257, 5)
    CRBM(4, 65)
    CRBM(256, 64, pooling=4)
    CRBM(100, 15)
    CRBM(20, 40)
    # alphabets other than DNA's (convRBM.py:84-87 warns and runs): up to 64 letters, letters x motif_length <= 2048
    with pytest.warns(UserWarning):
        m = CRBM(8, 12, input_dims=20)
    assert m.motifs.get_value().shape == (8, 1, 20, 12) and m.c.get_value().shape == (1, 20)
    with pytest.warns(UserWarning), pytest.raises(Exception, match="input_dims > 64"):
        CRBM(4, 5, input_dims=65)
    with pytest.warns(UserWarning), pytest.raises(Exception, match=r"input_dims \* motif_length > 2048"):
        CRBM(4, 103, input_dims=20)


def test_shape_checks_before_the_c_side_reads(monkeypatch):
    """set_fantasy / set_velocities hand raw pointers to C, which reads B*K*Lf floats."""
    from crbm_amd import CRBM
    m = CRBM(3, 4, batchsize=8, fantasy_hidden_len=10)
    monkeypatch.setattr(m, "_h", lambda: None)           # no GPU here: the checks must fire before any call
    monkeypatch.setattr(m, "_call", lambda *a: (_ for _ in ()).throw(AssertionError("reached the library")))
    with pytest.raises(ValueError, match="expected shape"):
        m.set_fantasy(np.zeros((4, 3, 1, 10), dtype=np.float32), np.zeros((4, 3, 1, 10), dtype=np.float32))
    with pytest.raises(ValueError, match="expected shape"):
        m.set_velocities(np.zeros((3, 1, 4, 5)), np.zeros((1, 3)), np.zeros((1, 4)))
    with pytest.raises(ValueError, match="hid_prime"):
        m.set_fantasy(np.zeros((8, 3, 1, 10), dtype=np.float32))


def test_load_state_defers_device_state(tmp_path):
    """loadState must leave the model attachable: nothing touches the GPU until the handle is made."""
    import joblib
    from crbm_amd import CRBM
    K, M, B, Lf = 3, 4, 8, 16
    rng = np.random.default_rng(0)
    fh = rng.binomial(1, 0.2, size=(B, K, 1, Lf)).astype(np.uint8)
    extra = {"velocities": (np.zeros((K, 1, 4, M), np.float32), np.zeros((1, K), np.float32), np.zeros((1, 4), np.float32)),
             "fantasy_bits": (np.packbits(fh, axis=3), np.packbits(fh[::-1], axis=3)), "fantasy_hidden_len": Lf,
             "world_size": 2, "rng": (77, 5, 1)}
    m0 = CRBM(K, M, batchsize=B, fantasy_hidden_len=Lf)
    params = (m0.motifs.get_value(), m0.bias.get_value(), m0.c.get_value())
    hyper = (K, M, 4, True, B, 0.1, 0.95, m0.rho, 0.1, 1, 5, 100, 'entropy')
    fn = str(tmp_path / "s.pkl")
    joblib.dump((params, hyper, extra), fn, protocol=2)
    m = CRBM.loadState(fn)
    assert m._handle is None and m._pending_state is not None and m.seed == 77
    got = []
    m.rank, m.world_size = 1, 2                          # as dist.attach would set them
    m.set_velocities = lambda *a: got.append(("v",))
    m.set_fantasy = lambda h, hp: got.append(("f", h, hp))
    m.set_rng = lambda *a: got.append(("r", a))
    m._install_state(m._pending_state)
    f = [g for g in got if g[0] == "f"][0]
    np.testing.assert_array_equal(f[1], fh[4:8].astype(np.float32))      # rank 1 of 2 takes chains 4..7
    np.testing.assert_array_equal(f[2], fh[::-1][4:8].astype(np.float32))
    assert [g for g in got if g[0] == "r"][0][1] == (77, 5, 1)


def test_bench_line_contract_of_the_tracked_profile():
    """The bench line the round's evidence holds (profiles/r04_bench_cfg2.json, written by `python bench.py` on an
    MI355X): the driver's keys, BASELINE.json's metric and config, a roofline object that names the resource that
    binds (no `bound: hbm`, no fraction above 1 posing as a utilisation) and is honest about partitioned launches
    (a step = launches_per_step overlapping launches; the step's time is not one launch's duration), the other
    BASELINE configurations on the same clock, and a CPU baseline on the cores the process may really use."""
    import json
    line = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_cfg2.json")).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"].split(";")[0].replace("x", "×") in base["metric"] and "cfg2" in line["config"]["workload"]
    assert line["vs_baseline"] is None and base["published"] == {} and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert abs(line["value"] - line["n_gpus"] * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]   # value from the host clock
    assert line["device_ms_per_step"] <= line["ms_per_step"] * 1.02 and line["metric_version"] == 3
    assert line["warmup_effective"] == line["warmup"] + line["chain_burn_in_launches"]
    r = line["roofline"]
    assert r["bound"] == "valu_issue" and 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["counters_source"].startswith("measured in this run")
    assert 0.0 < r["hbm_actual_frac"] < 0.2 and abs(r["hbm_actual_frac"] - r["traffic"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-6
    assert abs(r["algorithmic_frac"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 8e12) < 1e-6
    assert r["algorithmic_bytes_per_launch"] == 8 * (10 * 186 + 4 * 200) * 8192
    assert abs(r["avg_launch_us"] - 1e3 * line["device_ms_per_step"]) < 1e-6                       # HIP events over the timed region / steps
    # partitioned launches: two overlapping launches per step, each lasting LONGER than a step costs
    assert r["launches_per_step"] == line["launch"]["chain_parts"] == 2
    assert r["kernel_avg_us"] > r["avg_launch_us"] > r["kernel_avg_us"] / 2
    assert 1500 < r["shader_clock_mhz_timed_run"] < 2600 and abs(r["peak"] - r["avg_launch_us"] * r["shader_clock_mhz_timed_run"]) < 1e-3 * r["peak"]
    assert abs(r["traffic"] - 2 * r["state_bytes_per_launch"] / 2) < 0.12 * r["state_bytes_per_launch"]      # the packed state, once in and once out
    oc = line["other_configs"]
    for name in ("cfg4", "cfg5"):
        o = oc[name]
        assert name in o["workload"] and o["launches"] >= 20 and 0.0 < o["roofline"]["frac"] <= 1.0
        assert o["train"]["ms_per_train_step"] > o["device_ms_per_launch"] and "crbm_train_local" not in o["train"]["bound_note"]
    assert 0.0 < oc["cfg1"]["cfg1_fit_seconds_gpu"] < 1.0 and oc["cfg1"]["cfg1_status_line"].startswith("Epoch 0: FE=")
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and 1 <= c["cores"] <= c["hardware_threads_of_the_host"] and c["value"] > 0
    assert 0.5 <= c["parallel_efficiency"] <= 1.2                                                  # >= 0.5
    assert c["cfg1_fit_seconds"] > 10 * oc["cfg1"]["cfg1_fit_seconds_gpu"]
    assert line["train"]["ms_per_train_step"] > 3 * line["device_ms_per_step"] / 2 and "crbm_train_local" in line["train"]["bound_note"]
