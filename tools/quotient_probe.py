#!/usr/bin/env python3
"""On the GPU box: the kernels of the quotient-polynomial stage (p2hot_quotient_polys) and of the partial products at the C3
shape (2^20 gates, 80 routed wires, 2 challenges), timed with the library's own HIP-event profile (tooling).
    python tools/quotient_probe.py [log_n]"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from plonky2_amd import Engine                                                   # noqa: E402
from plonky2_amd.fri.oracle import PolynomialBatch                                # noqa: E402
from plonky2_amd.plonk.prover import all_wires_permutation_partial_products, compute_quotient_polys   # noqa: E402
from plonky2_amd.util.synthetic import splitmix_columns_torch                     # noqa: E402

P = 0xFFFFFFFF00000001
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(0)
dev = eng.mem.device
n, rb, cap = 1 << log_n, 3, 4
wires = splitmix_columns_torch(torch, dev, 0, 135, n)
cs = splitmix_columns_torch(torch, dev, 1000, 84, n)
k_is = [pow(14293326489335486720, j, P) for j in range(80)]
b_cs = PolynomialBatch.from_values(cs, rb, False, cap, engine=eng)
b_w = PolynomialBatch.from_values(wires, rb, False, cap, engine=eng)
zs = all_wires_permutation_partial_products(wires[:80], cs[4:84], k_is, 8, [3, 5], [11, 13], eng)
b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=eng)
for rep in range(3):
    torch.cuda.synchronize()
    eng.profile(True)
    chunks = compute_quotient_polys(b_w, b_cs, 4, b_z, k_is, 8, [3, 5], [11, 13], [17, 19], engine=eng)
    torch.cuda.synchronize()
    prof = eng.profile_results()
    eng.profile(False)
    del chunks
print(json.dumps({k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in prof.items()}, indent=1))
nq = n << 3
bytes_read = (80 + 80 + 2 * 20) * nq * 8
print("quotient_perm reads %.2f GB of LDE columns (wires 80, sigmas 80, Zs + partial products 20 twice)" % (bytes_read / 1e9))
