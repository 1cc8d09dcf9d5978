#!/usr/bin/env python3
"""On the GPU box: the library's own HIP-event profile of the NON-commit stages of the per-proof path at the C3 shape (partial
products, quotient polynomials, OpeningSet, prove_openings), one table per stage (tooling).
    python tools/path_probe.py [log_n]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from plonky2_amd import Engine                                                   # noqa: E402
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings   # noqa: E402
from plonky2_amd.iop.challenger import Challenger                                # noqa: E402
from plonky2_amd.plonk.prover import all_wires_permutation_partial_products, compute_quotient_polys   # noqa: E402
from plonky2_amd.util.synthetic import splitmix_columns_torch                     # noqa: E402

P = 0xFFFFFFFF00000001
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(0)
dev = eng.mem.device
n, rb, cap = 1 << log_n, 3, 4
arity = [4, 4, 4, 4] if log_n >= 16 else [4, 4]
wires = splitmix_columns_torch(torch, dev, 0, 135, n)
cs = splitmix_columns_torch(torch, dev, 1000, 84, n)
k_is = [pow(14293326489335486720, j, P) for j in range(80)]
b_cs = PolynomialBatch.from_values(cs, rb, False, cap, engine=eng)
b_w = PolynomialBatch.from_values(wires, rb, False, cap, engine=eng)


def staged(name, fn, reps=3):
    out = None
    for _ in range(reps):
        torch.cuda.synchronize()
        eng.profile(True)
        out = fn()
        torch.cuda.synchronize()
        prof = eng.profile_results()
        eng.profile(False)
    tot = sum(v["ms"] for v in prof.values())
    print("== %s: %.3f ms of kernels" % (name, tot))
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-28s %8.3f ms  x%d" % (k, v["ms"], v["launches"]))
    return out


zs = staged("partial products", lambda: all_wires_permutation_partial_products(wires[:80], cs[4:84], k_is, 8, [3, 5], [11, 13], eng))
b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=eng)
chunks = staged("quotient polys", lambda: compute_quotient_polys(b_w, b_cs, 4, b_z, k_is, 8, [3, 5], [11, 13], [17, 19], engine=eng))
b_q = PolynomialBatch.from_coeffs(chunks, rb, False, cap, engine=eng)
oracles = [b_cs, b_w, b_z, b_q]
widths = (84, 135, 20, 16)
allp = [(oi, pi) for oi, W in enumerate(widths) for pi in range(W)]
ch = Challenger(eng)
ch.observe_elements(np.arange(8, dtype=np.uint64))
zeta = ch.get_extension_challenge()
gz = [(zeta[0] * 7) % P, zeta[1]]
staged("OpeningSet", lambda: (eval_openings(oracles, [zeta], eng), eval_openings([b_z], [gz], eng)))


def po():
    c2 = Challenger(eng)
    c2.observe_elements(np.arange(8, dtype=np.uint64))
    c2.get_extension_challenge()
    return prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, [(2, 0), (2, 1)])], oracles, c2, rb, cap, arity, 16, 28, engine=eng)


staged("prove_openings", po)
