#!/usr/bin/env python3
"""PCIe-inclusive timing of the host-pointer entry point p2hot_commit (what the Rust shim calls): W separate host
vectors in, coefficients + digests + cap out, with and without the row-major leaf matrix (tooling).
usage: bench_hostapi.py [log_n] [W]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonky2_amd.util.synthetic import splitmix_columns_numpy  # noqa: E402
from plonky2_amd import Engine  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 135
rb, cap = 3, 4
eng = Engine(0)
n, N = 1 << log_n, 1 << (log_n + rb)
cols = [np.ascontiguousarray(c) for c in splitmix_columns_numpy(0, W, n)]
ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
coeffs = np.zeros((W, n), dtype=np.uint64)
nd = eng.num_digests(log_n + rb, cap)
digests = np.zeros((nd, 4), dtype=np.uint64)
capv = np.zeros((1 << cap, 4), dtype=np.uint64)
leaves = np.zeros((N, W), dtype=np.uint64)  # touched once here so page faults are not timed
for want_leaves in (False, True, False, True):
    t0 = time.perf_counter()
    eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data,
                                   leaves.ctypes.data if want_leaves else None, digests.ctypes.data, capv.ctypes.data, None))
    dt = time.perf_counter() - t0
    print("p2hot_commit W=%d 2^%d rows, leaves_out=%s: %.1f ms  -> %.2f GFE/s PCIe-inclusive"
          % (W, log_n, want_leaves, dt * 1e3, W * N / dt / 1e9))

# digests kept on the device (paths served from the handle by p2hot_batch_paths): only the columns go in and the
# coefficients + cap come back
for _ in range(2):
    handle = C.c_void_p()
    t0 = time.perf_counter()
    eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, None, None, capv.ctypes.data,
                                   C.byref(handle)))
    dt = time.perf_counter() - t0
    idx = np.arange(28, dtype=np.uint64) * 12345 % N
    paths = np.zeros((28, log_n + rb - cap, 4), dtype=np.uint64)
    t1 = time.perf_counter()
    eng.check(eng.lib.p2hot_batch_paths(handle, idx.ctypes.data, 28, paths.ctypes.data))
    dq = time.perf_counter() - t1
    eng.lib.p2hot_batch_free(handle)
    print("p2hot_commit W=%d 2^%d rows, digests on device: %.1f ms  -> %.2f GFE/s PCIe-inclusive; 28 paths %.2f ms"
          % (W, log_n, dt * 1e3, W * N / dt / 1e9, dq * 1e3))
