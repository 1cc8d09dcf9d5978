#!/usr/bin/env python3
"""Timeline of one host-pointer C3 wires commit (tooling): kernels and PCIe copies of the last p2hot_commit call, so the overlap of
the uploads, the transforms, the chunked leaf sponge and the result copies can be read off.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -o t -- python $REPO/tools/host_trace.py run
                 python $REPO/tools/host_trace.py report out"""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import ctypes as C
    import time
    import numpy as np
    from plonky2_amd import Engine
    from plonky2_amd.util.synthetic import splitmix_columns_numpy
    eng = Engine(0)
    W, log_n, rb, cap = 135, 20, 3, 4
    n = 1 << log_n
    cols = [np.ascontiguousarray(c) for c in splitmix_columns_numpy(0, W, n)]
    ptrs = (C.c_void_p * W)(*[c.ctypes.data for c in cols])
    coeffs = np.zeros((W, n), dtype=np.uint64)
    digests = np.zeros((eng.num_digests(log_n + rb, cap), 4), dtype=np.uint64)
    capv = np.zeros((1 << cap, 4), dtype=np.uint64)
    for it in range(3):
        t0 = time.perf_counter()
        h = C.c_void_p()
        eng.check(eng.lib.p2hot_commit(eng.ctx, ptrs, W, log_n, rb, cap, 1, 0, coeffs.ctypes.data, None, digests.ctypes.data,
                                       capv.ctypes.data, C.byref(h)))
        print("iteration", it, "p2hot_commit %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        eng.lib.p2hot_batch_free(h)
        time.sleep(0.05)  # a gap the report can cut at


def report(d):
    ev = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(\w+)\s*(<|\(|$)", r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0].split("::")[-1] + "(")
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", m.group(1) if m else r["Kernel_Name"][:40]))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kind = r.get("Direction", r.get("Name", "copy"))
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", kind[:40]))
    ev.sort()
    if not ev:
        print("no trace found under", d)
        return
    # the last call: everything after the last gap longer than 20 ms
    cut = 0
    for i in range(1, len(ev)):
        if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 20e6:
            cut = i
    ev = ev[cut:]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    print("last call: %d events, span %.2f ms" % (len(ev), (t1 - t0) / 1e6))

    def busy(sel):
        iv = sorted((a, b) for a, b, k, nme in ev if sel(k, nme))
        tot, cur_a, cur_b = 0, None, None
        for a, b in iv:
            if cur_b is None or a > cur_b:
                if cur_b is not None:
                    tot += cur_b - cur_a
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        if cur_b is not None:
            tot += cur_b - cur_a
        return tot / 1e6, (iv[0][0] - t0) / 1e6 if iv else 0, (max(b for _, b in iv) - t0) / 1e6 if iv else 0

    rows = [("kernels (any)", lambda k, n: k == "K"), ("  leaf sponge", lambda k, n: k == "K" and "hash_leaves" in n),
            ("  NTT passes", lambda k, n: k == "K" and ("limbpass" in n or "regpass" in n or "bitrev" in n)),
            ("  tree levels", lambda k, n: k == "K" and "merkle_level" in n),
            ("host-to-device copies", lambda k, n: k == "C" and ("HOST_TO_DEVICE" in n.upper() or "H2D" in n.upper())),
            ("device-to-host copies", lambda k, n: k == "C" and ("DEVICE_TO_HOST" in n.upper() or "D2H" in n.upper()))]
    for name, sel in rows:
        b, first, last = busy(sel)
        print("%-24s busy %7.2f ms   first start %7.2f ms   last end %7.2f ms" % (name, b, first, last))
    both = busy(lambda k, n: True)[0]
    print("union of everything      busy %7.2f ms" % both)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
