#!/usr/bin/env python3
"""Commit time vs P2HOT_NTT_ZLOOP_MIN (tooling): run once per value, prints ms for a few (W, log_n) shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plonky2_amd import Engine
from plonky2_amd.util.synthetic import splitmix_columns_torch
eng = Engine(0)
dev = torch.device("cuda:0")
out = []
for (W, k) in ((135, 12), (135, 14), (135, 16), (20, 16), (135, 18), (20, 20), (16, 20), (2, 20)):
    cols = splitmix_columns_torch(torch, dev, 0, W, 1 << k)
    f = lambda: eng.commit(cols, k, 3, 4, True)
    f(); torch.cuda.synchronize()
    reps = 20 if k <= 16 else 5
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    out.append("W%d k%d %.3f" % (W, k, (time.perf_counter() - t0) / reps * 1e3))
print("zloop_min=%s: %s" % (os.environ.get("P2HOT_NTT_ZLOOP_MIN", "default"), "  ".join(out)))
