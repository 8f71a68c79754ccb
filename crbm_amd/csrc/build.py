"""Builds libcrbm_hip.so (gfx950) in-tree with hipcc.

    python -m crbm_amd.csrc.build        # or: python crbm_amd/csrc/build.py

One object per instantiated motif-quad count (compiled in parallel), plus the
host API; everything links into crbm_amd/csrc/libcrbm_hip.so.  Objects are
rebuilt only when a source they depend on is newer.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcrbm_hip.so")
NQS = [1, 2, 3, 4, 5, 6, 8, 10, 13, 16]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
HEADERS = ["crbm_kernels.h", "crbm_layout.h", os.path.join("..", "..", "include", "crbm_amd.h")]


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
    for nq in NQS:
        work.append(("crbm_kernels_inst.hip", os.path.join(OBJ, "kernels_nq%d.o" % nq), ["-DCRBM_NQ=%d" % nq]))
    jobs = jobs or min(8, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(_compile, work))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build()
