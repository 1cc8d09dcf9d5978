#!/usr/bin/env python3
"""Kernel timeline of one p2hot_prove_openings_many call (tooling): run under rocprofv3 --kernel-trace, then summarise per stream.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python $REPO/tools/pom_trace.py run M
                 python $REPO/tools/pom_trace.py report out"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(M):
    import ctypes as C
    import numpy as np
    from plonky2_amd import Engine, _lib
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    eng = Engine(0)
    log_n, rb, cap = 12, 3, 4
    n = 1 << log_n
    widths = (135, 20, 16)
    cols = [splitmix_columns_numpy(b0, w, n) for b0, w in ((0, 135), (1000, 20), (2000, 16))]
    allp = [(oi, pi) for oi, w in enumerate(widths) for pi in range(w)]
    oi = (C.c_uint32 * len(allp))(*[o for o, _ in allp])
    pi = (C.c_uint32 * len(allp))(*[q for _, q in allp])
    arity = (C.c_uint * 2)(4, 4)
    fp = _lib.FriParams(rb, cap, 16, 28, arity, 2, 0, 0, 0)
    handles = []
    for cols_, w, isv in zip(cols, widths, (1, 1, 0)):
        ptrs = (C.c_void_p * (M * w))(*([cols_[e].ctypes.data for e in range(w)] * M))
        hs = (C.c_void_p * M)()
        eng.check(eng.lib.p2hot_commit_many(eng.ctx, ptrs, M, w, log_n, rb, cap, isv, None, None, None, hs))
        handles.append(hs)
    lay = _lib.FriProofLayout()
    h0 = (C.c_void_p * 3)(handles[0][0], handles[1][0], handles[2][0])
    eng.check(eng.lib.p2hot_fri_proof_sizes(h0, 3, C.byref(fp), C.byref(lay)))
    import time
    for it in range(3):
        chs = []
        for _ in range(M):
            h = C.c_void_p()
            eng.check(eng.lib.p2hot_challenger_create(eng.ctx, C.byref(h)))
            chs.append(h)
        bufs = [[np.zeros(max(1, getattr(lay, k + "_words")), dtype=np.uint64) for k in ("caps", "final_poly", "initial_leaves", "initial_paths", "step_evals", "step_paths")] for _ in range(M)]
        proofs = (_lib.FriProof * M)()
        infos = (_lib.FriBatchInfo * 1)()
        infos[0].point[0], infos[0].point[1] = 3, 5
        infos[0].oracle_index, infos[0].poly_index, infos[0].n_polys = oi, pi, len(allp)
        bp = (C.POINTER(_lib.FriBatchInfo) * M)(*([C.cast(infos, C.POINTER(_lib.FriBatchInfo))] * M))
        nb = (C.c_size_t * M)(*([1] * M))
        hs_all = (C.c_void_p * (3 * M))(*[handles[o][m] for m in range(M) for o in range(3)])
        for m in range(M):
            b = bufs[m]
            proofs[m] = _lib.FriProof(b[0].ctypes.data, b[1].ctypes.data, 0, None, b[2].ctypes.data, b[3].ctypes.data, b[4].ctypes.data, b[5].ctypes.data)
        cp = (C.c_void_p * M)(*chs)
        t0 = time.perf_counter()
        eng.check(eng.lib.p2hot_prove_openings_many(eng.ctx, M, bp, nb, hs_all, 3, cp, C.byref(fp), proofs))
        print("iteration", it, "M", M, "prove_openings_many %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)


def report(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if not rows:
        print("no kernel trace")
        return
    # the last third of the kernels = the last iteration, roughly
    rows = rows[len(rows) * 2 // 3:]
    t0 = int(rows[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    queues = {}
    for r in rows:
        q = r.get("Queue_Id", "?")
        queues.setdefault(q, [0, 0])
        queues[q][0] += 1
        queues[q][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("kernels %d  span %.3f ms  sum of kernel time %.3f ms  queues %s" % (len(rows), (t1 - t0) / 1e6, busy / 1e6,
                                                                          {q: (c, round(ns / 1e6, 3)) for q, (c, ns) in queues.items()}))
    # overlap: time during which >= 2 kernels run
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    depth, last, hist = 0, ev[0][0], {}
    for t, dlt in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += dlt
        last = t
    print("time at concurrency depth (ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
    names = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][-40:]
        names.setdefault(k, [0, 0])
        names[k][0] += 1
        names[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, ns) in sorted(names.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  %-42s %5d  %.3f ms" % (k, c, ns / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2])
