#!/bin/bash
# Development helper: builds tests/emu/libcrbm_emu.so exactly as tests/test_emu.py does.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
g++ -std=c++17 -O1 -g1 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -mf16c \
  -fPIC -shared -I "$R/tests/emu/shim" -I "$R/crbm_amd/csrc" "$R/tests/emu/emu_main.cpp" -o "$R/tests/emu/libcrbm_emu.so" -lpthread
echo built "$R/tests/emu/libcrbm_emu.so"
