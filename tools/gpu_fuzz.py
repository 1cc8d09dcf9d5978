#!/usr/bin/env python3
"""On the GPU box: time-boxed randomized comparison of the host-pointer ABI with the oracle on FRESH seeds and larger shapes
than the fixed-seed tests of tests/test_parity.py (tooling; the oracle is the checker, as in the tests).
    python tools/gpu_fuzz.py [seconds, default 240] [seed, default from the clock]
Shapes: W in 1..200, 2^0..2^15 rows, rate 1/1..1/16, any cap height, values / coefficients, salted or not, every output on or
off, kept values, forced column blocks (P2HOT_HOST_BLOCK_COLS is per context: a fresh context per block width), batched
commits (p2hot_commit_many), partial products + quotient polynomials on small instances.  Prints one line per failure and a
summary; exit code 1 on any mismatch."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import p2oracle as ora            # noqa: E402
from plonky2_amd import Engine, _lib           # noqa: E402

P = 0xFFFFFFFF00000001
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("gpu_fuzz seed", seed, flush=True)


def field(*shape):
    x = rng.integers(0, 2**64, size=shape, dtype=np.uint64)
    edge = rng.integers(0, 40, size=shape)
    x = np.where(edge == 0, np.uint64(P - 1), x)
    x = np.where(edge == 1, np.uint64(2**64 - 1), x)
    x = np.where(edge == 2, np.uint64(0), x)
    x = np.where(edge == 3, np.uint64(P), x)
    return np.ascontiguousarray(x)


engines = {}


def engine(block):
    if block not in engines:
        if block:
            os.environ["P2HOT_HOST_BLOCK_COLS"] = str(block)
        else:
            os.environ.pop("P2HOT_HOST_BLOCK_COLS", None)
        engines[block] = Engine(0)
    return engines[block]


fails = trials = 0
t_end = time.time() + budget
while time.time() < t_end:
    trials += 1
    W = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 84, 135, int(rng.integers(1, 200))]))
    log_n = int(rng.integers(0, 16 if W <= 20 else 13))
    rb = int(rng.integers(0, 5))
    cap = int(rng.integers(0, min(log_n + rb, 6) + 1))
    is_values = bool(rng.integers(0, 2))
    S = int(rng.choice([0, 0, 0, 4]))
    block = int(rng.choice([0, 0, 8, 16, 24]))
    eng = engine(block)
    n, N = 1 << log_n, 1 << (log_n + rb)
    cols = field(W, n)
    salts = field(S, N) if S else None
    o = ora.commit_salted(cols, salts, rb, cap, is_values) if S else ora.commit(cols, rb, cap, is_values)
    ptrs = (C.c_void_p * W)(*[cols[c].ctypes.data for c in range(W)])
    want_leaves, want_dig, want_co = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    LW = W + S
    coeffs = np.zeros((W, n), dtype=np.uint64)
    leaves = np.zeros((N, LW), dtype=np.uint64)
    nd = eng.num_digests(log_n + rb, cap)
    digests = np.zeros((max(nd, 1), 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    h = C.c_void_p()
    # round 5: the leaf matrix asynchronously (a pinned block, fenced before the comparison) and / or in natural LDE order
    lflags = int(rng.choice([0, 4, 8, 12])) if want_leaves else 0
    blk = C.c_void_p()
    if lflags & 4:
        eng.check(eng.lib.p2hot_host_alloc(eng.ctx, N * LW * 8, C.byref(blk)))
        leaves = np.frombuffer((C.c_uint64 * (N * LW)).from_address(blk.value), dtype=np.uint64).reshape(N, LW)
    args = [coeffs.ctypes.data if want_co else None, leaves.ctypes.data if want_leaves else None, digests.ctypes.data if want_dig else None,
            capv.ctypes.data, C.byref(h)]
    if S:
        sp = (C.c_void_p * S)(*[salts[j].ctypes.data for j in range(S)])
        rc = eng.lib.p2hot_commit_salted(eng.ctx, ptrs, W, log_n, rb, cap, 1 if is_values else 0, lflags, sp, S, *args)
    else:
        rc = eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1 if is_values else 0, lflags, *args)
    tag = dict(W=W, log_n=log_n, rb=rb, cap=cap, values=is_values, S=S, block=block, leaves=want_leaves, dig=want_dig, co=want_co, lflags=lflags)
    if rc == _lib.OK and lflags & 4:
        lo = int(rng.integers(0, N))
        eng.check(eng.lib.p2hot_batch_leaves_wait(h, lo, N))   # a sub-range first, then the rest
        eng.check(eng.lib.p2hot_batch_leaves_wait(h, 0, N))
    want = o["leaves"]
    if lflags & 8:
        bits = log_n + rb
        want = o["leaves"][np.array([int(format(i, "0%db" % bits)[::-1], 2) if bits else 0 for i in range(N)], dtype=np.int64)]
    bad = []
    if rc != _lib.OK:
        bad.append("rc %d %s" % (rc, eng.lib.p2hot_last_error(eng._ctx)))
    else:
        if (capv != o["cap"]).any():
            bad.append("cap")
        if want_co and (coeffs != o["coeffs"] % np.uint64(P)).any():
            bad.append("coeffs")
        if want_leaves and (leaves != want).any():
            bad.append("leaves")
        if want_dig and nd and (digests[:nd] != o["digests"]).any():
            bad.append("digests")
        idx = rng.integers(0, N, size=4).astype(np.uint64)
        rows = np.zeros((4, LW), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_rows(h, idx.ctypes.data, 4, rows.ctypes.data))
        if (rows != o["leaves"][idx.astype(np.int64)]).any():
            bad.append("rows")
        d2 = np.zeros((max(nd, 1), 4), dtype=np.uint64)
        eng.check(eng.lib.p2hot_batch_digests(h, d2.ctypes.data))
        if nd and (d2[:nd] != o["digests"]).any():
            bad.append("batch_digests")
        eng.lib.p2hot_batch_free(h)
    if blk.value:
        leaves = None
        eng.lib.p2hot_host_free(eng.ctx, blk)
    if bad:
        fails += 1
        print("MISMATCH", bad, tag, flush=True)
print("gpu_fuzz: %d trials, %d failures (seed %d)" % (trials, fails, seed))
sys.exit(1 if fails else 0)
