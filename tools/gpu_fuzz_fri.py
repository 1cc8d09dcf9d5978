#!/usr/bin/env python3
"""On the GPU box: time-boxed randomized opening proofs (commits, OpeningSet evaluations, p2hot_prove_openings) on random FRI
instances, each checked by the restated reference verifier, its wire-format round trip and five kinds of tampering -- the body
of tests/test_prove_openings.py::_fri_proof_verifies on shapes the fixed test list does not contain (tooling).
    python tools/gpu_fuzz_fri.py [seconds, default 240] [seed]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import p2oracle as ora                          # noqa: E402
from plonky2_amd import Engine                               # noqa: E402
from tests.test_prove_openings import _fri_proof_verifies    # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("gpu_fuzz_fri seed", seed, flush=True)
eng = Engine(0)
fails = trials = 0
t_end = time.time() + budget
while time.time() < t_end:
    trials += 1
    log_n = int(rng.integers(3, 12))   # (tiny domains make the tampering checks pass by chance: a changed PoW witness can land on the same query indices)
    n_or = int(rng.integers(1, 5))
    widths = [int(rng.integers(1, 24)) for _ in range(n_or)]
    rb = int(rng.integers(1, 4))
    if log_n + rb < 6:
        log_n = 6 - rb
    cap = int(rng.integers(0, min(log_n + rb, 5) + 1))
    arity, left = [], log_n
    while left > 0 and rng.integers(0, 4) != 0:
        a = int(rng.integers(1, min(4, left) + 1))
        # a round's tree has (current size >> a) leaves and needs at least 2^cap of them
        if (log_n + rb - sum(arity) - a) < cap:
            break
        arity.append(a)
        left -= a
    pow_bits = int(rng.integers(0, 9))
    nq = int(rng.integers(2, 7))
    blinded = tuple(oi for oi in range(n_or) if rng.integers(0, 5) == 0)
    nxt = None
    if rng.integers(0, 2):
        oi = int(rng.integers(0, n_or))
        nxt = (oi, int(rng.integers(1, widths[oi] + 1)))
    tag = dict(log_n=log_n, widths=widths, rb=rb, cap=cap, arity=arity, pow_bits=pow_bits, nq=nq, blinded=blinded, nxt=nxt)
    try:
        _fri_proof_verifies(eng, ora, log_n, widths, rb, cap, arity, pow_bits, nq, blinded, nxt)
    except BaseException:  # noqa: BLE001  (pytest.raises reports through a BaseException)
        fails += 1
        print("FAIL", tag, flush=True)
        traceback.print_exc(limit=3)
print("gpu_fuzz_fri: %d trials, %d failures (seed %d)" % (trials, fails, seed))
sys.exit(1 if fails else 0)
