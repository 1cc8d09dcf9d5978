#!/usr/bin/env python3
"""On the GPU box: time-boxed randomized sharded commits (p2hot_group_commit: `world` ranks of one process, all on device 0,
exchanging by copies) against the oracle -- coset, sub-coset (more ranks than LDE cosets) and column modes, random pipeline
chunking (tooling).
    python tools/gpu_fuzz_group.py [seconds, default 180] [seed]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import p2oracle as ora                       # noqa: E402
from plonky2_amd import Engine                            # noqa: E402
from plonky2_amd.distributed import GroupCommit           # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("gpu_fuzz_group seed", seed, flush=True)
eng = Engine(0)
groups = {w: GroupCommit(eng.lib, w, [0] * w) for w in (2, 4, 8)}
fails = trials = 0
t_end = time.time() + budget
while time.time() < t_end:
    trials += 1
    world = int(rng.choice([2, 4, 8]))
    lw = world.bit_length() - 1
    W = int(rng.choice([1, 2, 3, 7, 8, 9, 20, 33, 135, int(rng.integers(1, 60))]))
    rb = int(rng.integers(0, 4))
    log_n = int(rng.integers(max(0, lw - rb), 13 if W <= 33 else 10))
    cap = int(rng.integers(lw, min(log_n + rb, 6) + 1))
    is_values = bool(rng.integers(0, 2))
    by_columns = bool(rng.integers(0, 4) == 0)
    chunks = int(rng.integers(1, 9))
    tag = dict(world=world, W=W, log_n=log_n, rb=rb, cap=cap, values=is_values, columns=by_columns, chunks=chunks)
    try:
        cols = rng.integers(0, 2**64, size=(W, 1 << log_n), dtype=np.uint64)
        o = ora.commit(cols, rb, cap, is_values)
        r = groups[world].commit(cols, rb, cap, is_values, want_leaves=True, pipeline_chunks=chunks, by_columns=by_columns)
        ok = (r["coeffs"] == o["coeffs"] % np.uint64(0xFFFFFFFF00000001)).all() and (r["cap"] == o["cap"]).all() and \
            (r["digests"] == o["digests"]).all() and (r["leaves"] == o["leaves"]).all()
        N = 1 << (log_n + rb)
        xs = [0, N - 1] + [int(x) for x in rng.integers(0, N, 3)]
        rows, paths = r["open"](xs)
        for x, row, path in zip(xs, rows, paths):
            ok = ok and (row == o["leaves"][x]).all() and ora.merkle_verify(row, x, o["cap"], path)
        r["free"]()
        if not ok:
            raise AssertionError("mismatch")
    except BaseException:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, flush=True)
        traceback.print_exc(limit=3)
print("gpu_fuzz_group: %d trials, %d failures (seed %d)" % (trials, fails, seed))
sys.exit(1 if fails else 0)
