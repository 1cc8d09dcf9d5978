#!/usr/bin/env python3
"""Kernel timeline of the whole 2^12-gate per-proof path (tooling): run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/k12_trace.py
then  python tools/k12_trace.py --summarise <dir>  prints, for the LAST repetition, kernel time vs span per stage."""
import csv
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = 2**64 - 2**32 + 1

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
    # a repetition starts at the first ntt kernel after a prove_openings tail (merkle_paths / gather kernels)
    marks = [i for i, r in enumerate(rows) if name(r) == "query_indices_kernel"]
    seg = rows[marks[-2] + 1:] if len(marks) >= 2 else rows
    # drop the tail of the previous repetition: start at the first NTT kernel
    first = next(i for i, r in enumerate(seg) if "ntt_regpass" in r["Kernel_Name"])
    seg = seg[first:]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("last repetition: %d kernels, span %.3f ms, kernel time %.3f ms, gaps %.3f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    per = {}
    for r in seg:
        d = per.setdefault(name(r), [0, 0])
        d[0] += 1
        d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, ns) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
        print("  %-36s x%-4d %8.3f ms" % (k, c, ns / 1e6))
    gaps = sorted(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3, name(a), name(b)) for a, b in zip(seg, seg[1:]))
    print("largest gaps (us):", [(round(g, 1), a, b) for g, a, b in gaps[-8:]])
    sys.exit(0)

import torch  # noqa: E402

from plonky2_amd import Engine  # noqa: E402
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings  # noqa: E402
from plonky2_amd.iop.challenger import Challenger  # noqa: E402
from plonky2_amd.plonk.prover import all_wires_permutation_partial_products  # noqa: E402
from plonky2_amd.util.synthetic import splitmix_columns_torch  # noqa: E402

eng = Engine(0)
dev = torch.device("cuda:0")
host = "--host" in sys.argv   # inputs as host arrays: every stage through the host-pointer entry points the Rust shim calls
if host:
    sys.argv.remove("--host")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n, rb, cap = 1 << log_n, 3, 4
arity = [4] * ((log_n - 4) // 4)
wires = splitmix_columns_torch(torch, dev, 0, 135, n)
sig = splitmix_columns_torch(torch, dev, 1000, 80, n)
quo = splitmix_columns_torch(torch, dev, 2000, 16, n)
k_is = [pow(14293326489335486720, j, P) for j in range(80)]
if host:
    wires, sig, quo = (eng.host(x) for x in (wires, sig, quo))
for rep in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b_w = PolynomialBatch.from_values(wires, rb, False, cap, engine=eng)
    zs = all_wires_permutation_partial_products(wires[:80], sig, k_is, 8, [3, 5], [11, 13], eng)
    b_z = PolynomialBatch.from_values(zs, rb, False, cap, engine=eng)
    b_q = PolynomialBatch.from_coeffs(quo, rb, False, cap, engine=eng)
    oracles = [b_w, b_z, b_q]
    ch = Challenger(eng)
    ch.observe_elements(np.arange(8, dtype=np.uint64))
    zeta = ch.get_extension_challenge()
    gz = [(zeta[0] * 7) % P, zeta[1]]
    eval_openings(oracles, [zeta, gz], eng)
    allp = [(oi, pi) for oi, W in enumerate((135, 20, 16)) for pi in range(W)]
    nxt = [(oi, pi) for oi, W in enumerate((135, 20)) for pi in range(W)]
    prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)], oracles, ch, rb, cap, arity, 16, 28, engine=eng)
    torch.cuda.synchronize()
    print("path 2^%d%s: %.3f ms" % (log_n, " (host pointers)" if host else "", (time.perf_counter() - t0) * 1e3))
