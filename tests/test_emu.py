"""Runs the HIP kernels on CPU threads under AddressSanitizer + UBSan
(tests/emu/) and checks each of them against the float64 oracle.  This is the
"sanitizers on the CPU build" leg: GPU ASan is not available on the pool.
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "emu")
CSRC = os.path.join(os.path.dirname(HERE), "crbm_amd", "csrc")
LIB = os.path.join(EMU, "libcrbm_emu.so")
SOURCES = [os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "shim", "hip", "hip_runtime.h"),
           os.path.join(CSRC, "crbm_kernels.h"), os.path.join(CSRC, "crbm_kernels_generic.h"), os.path.join(CSRC, "crbm_layout.h")]


def _gcc_file(name):
    return subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()


@pytest.fixture(scope="module")
def emu_env():
    if not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in SOURCES):
        cmd = ["g++", "-std=c++17", "-O1", "-g1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
               "-fno-sanitize-recover=undefined", "-mf16c", "-fPIC", "-shared", "-I", os.path.join(EMU, "shim"), "-I", CSRC,
               os.path.join(EMU, "emu_main.cpp"), "-o", LIB, "-lpthread"]
        subprocess.check_call(cmd)
    env = dict(os.environ)
    env["LD_PRELOAD"] = _gcc_file("libasan.so") + ":" + _gcc_file("libubsan.so")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    return env


CASES = ["encode_pack", "hgv", "vgh", "gibbs", "stats_mfma", "train_step", "two_ranks", "pooling",
         "free_energy", "hit_summary", "large_models", "big", "slabs"]


@pytest.fixture(scope="module")
def emu_runs(emu_env):
    """All harness cases, four at a time (each is a process full of mostly sleeping threads: the kernels' GPU threads
    are OS threads here); the tests below each look at one result."""
    from concurrent.futures import ThreadPoolExecutor

    def run(which):
        try:
            return subprocess.run([sys.executable, os.path.join(EMU, "run_emu.py"), which], env=emu_env,
                                  capture_output=True, text=True, timeout=1500)
        except subprocess.TimeoutExpired as e:
            return e
    order = sorted(CASES, key=lambda w: {"big": 0, "large_models": 1, "hit_summary": 2}.get(w, 3))   # longest first
    with ThreadPoolExecutor(max_workers=4) as ex:
        return dict(zip(order, ex.map(run, order)))


@pytest.mark.parametrize("which", CASES)
def test_kernels_on_cpu_threads_with_sanitizers(emu_runs, which):
    r = emu_runs[which]
    assert not isinstance(r, subprocess.TimeoutExpired), "emulator case %s timed out" % which
    assert r.returncode == 0 and "EMU ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
