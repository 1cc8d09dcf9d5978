#!/usr/bin/env python3
"""PCIe-inclusive p2hot_commit (C3 wires from pageable host vectors) with the chunked leaf sponge on and off (tooling).
usage: hostapi_ab.py [log_n] [W] [leaves]      (leaves: also copy the row-major leaf matrix back, P2HOT_HOST_LEAVES_FIRST on / off)"""
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
n, N = 1 << log_n, 1 << (log_n + rb)
cols = [np.ascontiguousarray(c) for c in splitmix_columns_numpy(0, W, n)]
ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
coeffs = np.zeros((W, n), dtype=np.uint64)
caps = {}
if len(sys.argv) > 3 and sys.argv[3] == "leaves":
    leaves = np.zeros((N, W), dtype=np.uint64)
    for mode in ("1", "0", "1", "0"):
        os.environ["P2HOT_HOST_LEAVES_FIRST"] = mode
        eng = Engine(0)
        digests = np.zeros((eng.num_digests(log_n + rb, cap), 4), dtype=np.uint64)
        capv = np.zeros((1 << cap, 4), dtype=np.uint64)
        ts = []
        for it in range(4):
            t0 = time.perf_counter()
            h = C.c_void_p()
            eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, leaves.ctypes.data,
                                           digests.ctypes.data, capv.ctypes.data, C.byref(h)))
            ts.append((time.perf_counter() - t0) * 1e3)
            eng.lib.p2hot_batch_free(h)
        caps.setdefault("cap", capv.copy())
        caps.setdefault("sum", int(leaves[::4097].sum() % (1 << 61)))
        assert (capv == caps["cap"]).all() and int(leaves[::4097].sum() % (1 << 61)) == caps["sum"]
        print("leaves_first=%s  ms: %s  (min %.2f)" % (mode, " ".join("%.1f" % t for t in ts), min(ts)))
        eng.close()
    sys.exit(0)
for mode in ("1", "0", "1", "0"):
    os.environ["P2HOT_HOST_CHUNKED_HASH"] = mode
    eng = Engine(0)
    nd = eng.num_digests(log_n + rb, cap)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    for want_dig in (True, False):
        ts = []
        for it in range(5):
            t0 = time.perf_counter()
            h = C.c_void_p()
            eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, None,
                                           digests.ctypes.data if want_dig else None, capv.ctypes.data, C.byref(h)))
            ts.append((time.perf_counter() - t0) * 1e3)
            eng.lib.p2hot_batch_free(h)
        caps.setdefault("cap", capv.copy())
        assert (capv == caps["cap"]).all()
        print("chunked=%s digests_out=%s  ms: %s  (min %.2f)" % (mode, want_dig, " ".join("%.1f" % t for t in ts), min(ts)))
    eng.close()
