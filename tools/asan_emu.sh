#!/bin/bash
# Runs the CPU-tier parity tests with the kernel-source emulator built under AddressSanitizer: "device" memory is
# malloc'ed there, so any out-of-bounds index in a kernel shows up as an ASAN report (tooling, test infrastructure).
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -DP2HOT_EMU -DEMU_UCONTEXT -Itests/emu -Iplonky2_amd/csrc \
    -shared -o /tmp/libp2hot_asan.so tests/emu/hip_emu.cpp tests/emu/gcn_asm.cpp -x c++ plonky2_amd/csrc/p2hot.hip
cat > /tmp/asan_run.py <<'P'
import sys
sys.path.insert(0, ".")
import tests.emu_backend as eb
from plonky2_amd import _lib
eb._lib_cache = _lib.load("/tmp/libp2hot_asan.so")
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "tests/test_parity.py", "tests/test_asm_streams.py", "tests/test_prove_openings.py", "tests/test_permutation.py", "tests/test_distributed.py", "-k", "(emu or group or asm or checker) and not c2_wires_golden", "-p", "no:cacheprovider"]))
P
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    python /tmp/asan_run.py 2>&1 | grep -v "doesn't fully support makecontext"
