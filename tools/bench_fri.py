#!/usr/bin/env python3
"""Times the FRI commit phase (p2hot_fri_commit, host pointers: includes H2D of the coefficients and
D2H of leaves/digests/caps) and the PoW grind at a BASELINE config.  usage: bench_fri.py [log_n rate_bits]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plonky2_amd import Engine
from plonky2_amd.fri.prover import fri_committed_trees, fri_proof_of_work
from plonky2_amd.iop.challenger import Challenger

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = Engine(0)
rng = np.random.default_rng(1)
co = rng.integers(0, 0xFFFFFFFF00000001, size=(1 << log_n, 2), dtype=np.uint64)
arity = [4, 4, 4, 4]
for rep in range(3):
    ch = Challenger(eng)
    eng.profile(True); eng.profile_results(reset=True)
    t0 = time.perf_counter()
    trees, final, betas = fri_committed_trees(co, ch, rb, 4, arity, engine=eng)
    t1 = time.perf_counter()
    w = fri_proof_of_work(ch, 16, engine=eng)
    t2 = time.perf_counter()
    prof = eng.profile_results(reset=True)
    print("fri_commit log_n=%d rate_bits=%d: %.2f ms (incl. PCIe both ways), pow(16 bits) %.2f ms witness %d; device kernels: %s"
          % (log_n, rb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, w, {k: round(v["ms"], 3) for k, v in prof.items()}))
