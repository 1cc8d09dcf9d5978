#!/usr/bin/env python3
"""On the GPU box: time-boxed randomized permutation-argument instances -- p2hot_partial_products and p2hot_quotient_polys against
the oracle's restatements, on shapes the fixed test lists do not contain (the bodies of tests/test_permutation.py) (tooling).
    python tools/gpu_fuzz_plonk.py [seconds, default 180] [seed]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import p2oracle as ora                                                        # noqa: E402
from plonky2_amd import Engine                                                             # noqa: E402
from tests.test_permutation import test_partial_products_vs_oracle, test_quotient_polys_vs_oracle   # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = np.random.default_rng(seed)
print("gpu_fuzz_plonk seed", seed, flush=True)
eng = Engine(0)
fails = trials = 0
t_end = time.time() + budget
while time.time() < t_end:
    trials += 1
    num_routed = int(rng.choice([3, 5, 8, 9, 16, 17, 24, 33, 80, int(rng.integers(3, 100))]))
    degree = int(rng.choice([8, 8, int(rng.integers(2, min(num_routed, 10)))]))
    if degree >= num_routed:
        degree = num_routed - 1
    qbits = (degree - 1).bit_length()
    rb = int(rng.integers(qbits, 4)) if qbits <= 3 else qbits
    nc = int(rng.choice([1, 2, 2, 2, 3, 4]))
    log_n = int(rng.integers(0, 9 if num_routed <= 33 else 7))
    tag = dict(num_routed=num_routed, degree=degree, log_n=log_n, rb=rb, nc=nc)
    try:
        test_partial_products_vs_oracle(eng, ora, num_routed, degree, log_n, nc)
        test_quotient_polys_vs_oracle(eng, ora, num_routed, degree, max(log_n, 2), rb, nc, bool(rng.integers(0, 2)))   # (the test body spoils row 2 of a wire: at least four rows)
    except BaseException:  # noqa: BLE001
        fails += 1
        print("FAIL", tag, flush=True)
        traceback.print_exc(limit=4)
print("gpu_fuzz_plonk: %d trials, %d failures (seed %d)" % (trials, fails, seed))
sys.exit(1 if fails else 0)
