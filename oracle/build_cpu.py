"""Builds oracle/libcrbm_cpu.so (the C restatement used as the CPU baseline).
TEST INFRASTRUCTURE: called by __graft_entry__.build(), tests and bench.py."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "crbm_cpu.c")
LIB = os.path.join(HERE, "libcrbm_cpu.so")


def build(verbose=False):
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    # no -march=native: the .so is built here and travels to a different host
    cmd = ["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-fPIC", "-shared", "-std=c11", SRC, "-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(True)
