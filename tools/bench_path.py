#!/usr/bin/env python3
"""Per-proof path timing on one MI355X (tooling; the contract bench is bench.py): the three PolynomialBatch commits
of a 2^log_n-gate standard_recursion_config proof (wires W=135 from_values, Zs+partial products W=20 from_values,
quotient chunks W=16 from_coeffs), OpeningSet evaluation, and prove_openings (final_poly over the 171 polynomials at
zeta plus the 135+20... opened at g*zeta in plonky2; here: every polynomial at zeta, wires+Zs at g*zeta), FRI commit,
PoW, 28 queries.  Inputs resident in HBM; times are host wall-clock around synchronised calls.
usage: bench_path.py [plonky2|starky] [log_n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from plonky2_amd.util.synthetic import splitmix_columns_torch  # noqa: E402
from plonky2_amd import Engine  # noqa: E402
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings  # noqa: E402
from plonky2_amd.iop.challenger import Challenger  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plonky2"
eng = Engine(0)
if mode == "plonky2":
    log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    shapes, rb, cap, arity, nq = [(135, True), (20, True), (16, False)], 3, 4, [4] * ((log_n - 4) // 4), 28
else:  # starky Fibonacci: trace W=2 (from_values) and quotient W=2 (from_coeffs), rate 1/2, 84 queries
    log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
    shapes, rb, cap, arity, nq = [(2, True), (2, False)], 1, 4, [4, 4, 4, 4], 84
n = 1 << log_n


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print("  %-44s %9.2f ms" % (label, dt))
    return r, dt


for rep in range(2):
    print("rep %d  (%s, 2^%d rows, rate 1/%d)" % (rep, mode, log_n, 1 << rb))
    total = 0.0
    oracles = []
    c0 = 0
    wires = None
    for (W, is_values) in shapes:
        if mode == "plonky2" and W == 20:
            # the real Zs + partial products of the first 80 wire columns (prover.rs:219-229); sigmas: any columns
            from plonky2_amd.plonk.prover import all_wires_permutation_partial_products
            sig = splitmix_columns_torch(torch, eng.mem.device, 1000, 80, n)
            k_is = [pow(14293326489335486720, j, 0xFFFFFFFF00000001) for j in range(80)]
            cols, dt = timed("partial products + Zs (80 routed wires, 2 challenges)",
                             lambda: all_wires_permutation_partial_products(wires[:80], sig, k_is, 8, [3, 5], [11, 13], eng))
            total += dt
        else:
            cols = splitmix_columns_torch(torch, eng.mem.device, c0, W, n)
        if wires is None:
            wires = cols
        c0 += W
        ctor = PolynomialBatch.from_values if is_values else PolynomialBatch.from_coeffs
        b, dt = timed("%s W=%d" % ("from_values" if is_values else "from_coeffs", W),
                      lambda: ctor(cols, rb, False, cap, engine=eng))
        total += dt
        oracles.append(b)
    ch = Challenger(eng)
    ch.observe_elements(np.arange(8, dtype=np.uint64))
    zeta = ch.get_extension_challenge()
    gz = [(zeta[0] * 7) % 0xFFFFFFFF00000001, zeta[1]]
    _, dt = timed("OpeningSet: all polys at 2 points", lambda: eval_openings(oracles, [zeta, gz], eng))
    total += dt
    allp = [(oi, pi) for oi, (W, _) in enumerate(shapes) for pi in range(W)]
    nxt = [(oi, pi) for oi, (W, _) in enumerate(shapes[:-1]) for pi in range(W)]
    batches = [FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)]
    tm = {}
    _, dt = timed("prove_openings (final_poly, FRI commit, PoW, %d queries)" % nq,
                  lambda: prove_openings(batches, oracles, ch, rb, cap, arity, 16, nq, engine=eng, timing=tm))
    for k, v in tm.items():
        print("      %-52s %7.2f ms" % (k, v))
    total += dt
    print("  %-44s %9.2f ms" % ("path total", total))
