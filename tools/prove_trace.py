#!/usr/bin/env python3
"""Kernel timeline of p2hot_prove_openings at recursion size (tooling): run under
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/prove_trace.py [log_n]
and summarise with tools/prove_trace.py --summarise <dir>: kernel time vs wall time of the last call, per-kernel totals."""
import csv
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call = the kernels after the last "alpha_powers_kernel"
    starts = [i for i, r in enumerate(rows) if "alpha_powers" in r["Kernel_Name"]]
    seg = rows[starts[-1] - 1:] if starts else rows
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    per = {}
    for r in seg:
        k = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
        d = per.setdefault(k, [0, 0])
        d[0] += 1
        d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("last prove_openings: %d kernels, span %.3f ms, kernel time %.3f ms, gaps %.3f ms"
          % (len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
    for k, (c, ns) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-34s x%-4d %8.3f ms" % (k, c, ns / 1e6))
    sys.exit(0)

import torch  # noqa: E402

from plonky2_amd import Engine  # noqa: E402
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, prove_openings  # noqa: E402
from plonky2_amd.iop.challenger import Challenger  # noqa: E402
from plonky2_amd.util.synthetic import splitmix_columns_torch  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
eng = Engine(0)
n, rb, cap = 1 << log_n, 3, 4
arity = [4] * ((log_n - 4) // 4)
oracles = [PolynomialBatch.from_values(splitmix_columns_torch(torch, eng.mem.device, c0, W, n), rb, False, cap, engine=eng)
           for c0, W in ((0, 135), (200, 20))] + \
          [PolynomialBatch.from_coeffs(splitmix_columns_torch(torch, eng.mem.device, 300, 16, n), rb, False, cap, engine=eng)]
allp = [(oi, pi) for oi, W in enumerate((135, 20, 16)) for pi in range(W)]
nxt = [(oi, pi) for oi, W in enumerate((135, 20)) for pi in range(W)]
for rep in range(4):
    ch = Challenger(eng)
    ch.observe_elements(np.arange(8, dtype=np.uint64))
    zeta = ch.get_extension_challenge()
    gz = [(zeta[0] * 7) % 0xFFFFFFFF00000001, zeta[1]]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prove_openings([FriBatchInfo(zeta, allp), FriBatchInfo(gz, nxt)], oracles, ch, rb, cap, arity, 16, 28, engine=eng)
    print("prove_openings 2^%d: %.3f ms" % (log_n, (time.perf_counter() - t0) * 1e3))
