#!/usr/bin/env python3
"""Tooling: commit time vs the quad-cooperative Poseidon threshold (p2hot_tune_quad) for small batches."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import splitmix_columns_torch  # noqa: E402
from plonky2_amd import Engine  # noqa: E402

eng = Engine(0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 135
for log_n in (10, 12, 13, 14, 15, 16):
    cols = splitmix_columns_torch(torch, eng.mem.device, 0, W, 1 << log_n)
    row = []
    for thr_log in (0, 15, 16, 17, 18, 19, 20):
        eng.check(eng.lib.p2hot_tune_quad(eng.ctx, (1 << thr_log) if thr_log else 0))
        for _ in range(3):
            eng.commit(cols, log_n, 3, 4, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.commit(cols, log_n, 3, 4, True)
        torch.cuda.synchronize()
        row.append("%d:%.3f" % (thr_log, (time.perf_counter() - t0) / 20 * 1e3))
    print("W=%d rows 2^%d (leaves 2^%d)  thr_log:ms  %s" % (W, log_n, log_n + 3, "  ".join(row)))
