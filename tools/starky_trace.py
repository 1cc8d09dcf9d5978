#!/usr/bin/env python3
"""Kernel timeline of the starky per-proof path (C4: W = 2, 2^22 rows, rate 1/2, 84 queries) (tooling): run under
rocprofv3 --kernel-trace, then `--summarise <dir>` lists the kernels of the last repetition by total time."""
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
    marks = [i for i, r in enumerate(rows) if name(r) == "query_indices_kernel"]
    seg = rows[marks[-2] + 1:]
    first = next(i for i, r in enumerate(seg) if "ntt_regpass" in r["Kernel_Name"])
    seg = seg[first:]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    print("last repetition: %d kernels, span %.3f ms, kernel time %.3f ms" % (len(seg), (t1 - t0) / 1e6, busy / 1e6))
    per = {}
    for r in seg:
        d = per.setdefault(name(r), [0, 0])
        d[0] += 1
        d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, ns) in sorted(per.items(), key=lambda kv: -kv[1][1])[:22]:
        print("  %-36s x%-4d %8.3f ms" % (k, c, ns / 1e6))
    prev = t0
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - prev) / 1e3
        prev = int(r["End_Timestamp"])
        if d > 100 or gap > 40:
            print("%9.1f us gap %6.1f +%8.1f %s grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, gap, d, name(r)[:40], r.get("Grid_Size_X")))
    sys.exit(0)

import torch  # noqa: E402

from plonky2_amd import Engine  # noqa: E402
from plonky2_amd.fri.oracle import FriBatchInfo, PolynomialBatch, eval_openings, prove_openings  # noqa: E402
from plonky2_amd.iop.challenger import Challenger  # noqa: E402
from plonky2_amd.util.synthetic import fibonacci_trace, splitmix_columns_torch  # noqa: E402

eng = Engine(0)
dev = torch.device("cuda:0")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n, rb, cap, nq, arity = 1 << log_n, 1, 4, 84, [4, 4, 4, 4]
trace = eng.dev(fibonacci_trace(log_n))
quo = splitmix_columns_torch(torch, dev, 3000, 2, n)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b_t = PolynomialBatch.from_values(trace, rb, False, cap, engine=eng)
    b_q = PolynomialBatch.from_coeffs(quo, rb, False, cap, engine=eng)
    ch = Challenger(eng)
    ch.observe_elements(np.arange(8, dtype=np.uint64))
    zeta = ch.get_extension_challenge()
    gz = [(zeta[0] * 7) % P, zeta[1]]
    eval_openings([b_t, b_q], [zeta, gz], eng)
    prove_openings([FriBatchInfo(zeta, [(0, 0), (0, 1), (1, 0), (1, 1)]), FriBatchInfo(gz, [(0, 0), (0, 1)])], [b_t, b_q], ch,
                   rb, cap, arity, 16, nq, engine=eng)
    torch.cuda.synchronize()
    print("starky path 2^%d: %.3f ms" % (log_n, (time.perf_counter() - t0) * 1e3))
