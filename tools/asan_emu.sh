#!/bin/bash
# Runs the CPU-tier parity tests with the kernel-source emulator built under AddressSanitizer (tooling, test infrastructure).
# Device memory is mmap'ed behind guard pages since round 4 (tests/emu/hip_emu_rt.cpp: an index off the END of a device buffer
# faults by itself); what ASAN adds is the HOST side -- the library's vectors, staging blocks and handles, the emulated runtime's
# queues, the fake RCCL -- and underruns / interior overruns into other heap objects.
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=c++17 -fPIC -DP2HOT_EMU -DEMU_UCONTEXT -Itests/emu -Iplonky2_amd/csrc \
    -shared -o /tmp/libp2hot_asan.so tests/emu/hip_emu.cpp tests/emu/hip_emu_rt.cpp tests/emu/gcn_asm.cpp -x c++ plonky2_amd/csrc/p2hot.hip -lrt
cat > /tmp/asan_run.py <<'P'
import sys
sys.path.insert(0, ".")
import tests.emu_backend as eb
from plonky2_amd import _lib
eb._lib_cache = _lib.load("/tmp/libp2hot_asan.so")
eb._SO = "/tmp/libp2hot_asan.so"   # (the memory backend looks the runtime's C++ symbols up in this file)
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "not gpu", "tests/test_parity.py", "tests/test_asm_streams.py", "tests/test_prove_openings.py", "tests/test_permutation.py", "tests/test_distributed.py", "tests/test_emu_devices.py", "tests/test_async_leaves.py", "tests/test_alloc_failures.py", "tests/test_proof_path.py", "-k", "(emu or group or asm or checker or devices or rccl or queues or waits or allocation or fence) and not c2_wires_golden and not gloo and not named_fault", "-p", "no:cacheprovider"]))
P
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    python /tmp/asan_run.py 2>&1 | grep -v "doesn't fully support makecontext"
