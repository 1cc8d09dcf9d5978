#!/usr/bin/env python3
"""Summaries of tools/profile_all.sh's rocprofv3 output (runs on the GPU box; tooling).

  <tag>_kernel_stats.csv   the --kernel-trace --stats table (our kernels only)
  <tag>_pmc_traffic.json   per kernel INSTANTIATION (template arguments kept): launches, HBM bytes per launch from the
                           FETCH_SIZE / WRITE_SIZE passes (KiB; FETCH doubled on gfx950 as MI355X_MICROARCH.md prescribes),
                           SQ_INSTS_VALU per launch, the measured clock, stamped with the hash of plonky2_amd/csrc
  <tag>_pmc_sq.txt         every SQ counter of the two SQ passes per kernel instantiation
usage: tools/prof_summarize.py <tag>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.csrc_hash import csrc_hash  # noqa: E402

G = os.path.join(ROOT, "gpurun_out")
OURS = ("merkle::", "ntt::", "nttl::", "fri::", "plonk::", "poseidon")


def short(name):
    """'void nttl::ntt_limbpass_kernel<false, 12, 0, 0, 0>(nttl::LimbPassArgs)' -> 'ntt_limbpass_kernel<false,12,0,0,0>'"""
    n = re.sub(r"^void ", "", name)
    n = n[:n.rfind("(")] if "(" in n else n
    m = re.match(r"(?:\w+::)*(\w+)(<.*>)?$", n)
    if not m:
        return n
    targs = (m.group(2) or "").replace(" ", "").replace("merkle::", "")
    return m.group(1) + targs


def counters(d):
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    dur = defaultdict(float)
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if not any(o in r["Kernel_Name"] for o in OURS):
                continue
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return acc, {k: len(v) for k, v in launches.items()}, dur


def main(tag):
    # kernel trace
    stats = glob.glob(os.path.join(G, "prof", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        lines = open(stats[0]).readlines()
        with open(os.path.join(G, "%s_kernel_stats.csv" % tag), "w") as f:
            f.writelines(lines[:1] + [l for l in lines[1:] if any(o in l for o in OURS)])
    fe, nf, _ = counters(os.path.join(G, "pmc_fetch"))
    wr, nw, _ = counters(os.path.join(G, "pmc_write"))
    sq, ns, ms = counters(os.path.join(G, "pmc_sq"))
    sq2, ns2, ms2 = counters(os.path.join(G, "pmc_sq2"))
    sq3, ns3, ms3 = counters(os.path.join(G, "pmc_sq3"))
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        n = max(nf.get(k, 0), nw.get(k, 0), 1)
        e = {"launches": n, "fetch_size_kib_raw": fe[k].get("FETCH_SIZE", 0.0), "write_size_kib_raw": wr[k].get("WRITE_SIZE", 0.0)}
        e["fetch_bytes_per_launch"] = 2.0 * e["fetch_size_kib_raw"] * 1024 / n  # gfx950: 128-B reads tallied at 64 B
        e["write_bytes_per_launch"] = e["write_size_kib_raw"] * 1024 / n
        e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
        if k in sq and ns.get(k):
            e["sq_insts_valu_per_launch"] = sq[k].get("SQ_INSTS_VALU", 0.0) / ns[k]
            if ms.get(k):
                e["ms_per_launch_under_pmc"] = ms[k] / ns[k]
                e["clock_ghz"] = sq[k].get("GRBM_GUI_ACTIVE", 0.0) / 8 / (ms[k] * 1e6)  # summed over the 8 XCDs
        if k in sq3 and sq3[k].get("GRBM_GUI_ACTIVE"):
            # VALU busy fraction: quad-cycles some wave had a VALU instruction executing / SIMD cycles (1024 SIMDs)
            cyc = sq3[k]["GRBM_GUI_ACTIVE"] / 8
            e["valu_busy_frac"] = sq3[k].get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024 * cyc)
            e["lds_busy_frac"] = sq3[k].get("SQ_ACTIVE_INST_LDS", 0.0) * 4 / (1024 * cyc)
            e["clock_ghz_sq3"] = cyc / (ms3[k] * 1e6) if ms3.get(k) else 0.0
        kernels[k] = e
    out = {"workload": {"W": 135, "log_n": 20, "rate_bits": 3, "cap_height": 4, "n_gpus": 1}, "csrc_sha256_16": csrc_hash(), "tag": tag,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `bench.py --steps 1 --warmup 0`; "
                     "FETCH_SIZE doubled (gfx950 tallies 128-B reads at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as "
                     "reported; KiB -> bytes.  SQ_INSTS_VALU / GRBM_GUI_ACTIVE from a third pass.  One entry per kernel instantiation "
                     "(tools/prof_summarize.py).",
           "kernels": kernels}
    json.dump(out, open(os.path.join(G, "%s_pmc_traffic.json" % tag), "w"), indent=1)
    with open(os.path.join(G, "%s_pmc_sq.txt" % tag), "w") as f:
        for k in sorted(set(sq) | set(sq2) | set(sq3)):
            c = dict(sq.get(k, {}))
            c.update(sq2.get(k, {}))
            c.update({a: b for a, b in sq3.get(k, {}).items() if a not in c})
            f.write("%s launches %d ms %.6f %s\n" % (k, ns.get(k, ns2.get(k, 0)), ms.get(k, ms2.get(k, 0.0)), {a: "%.4g" % b for a, b in sorted(c.items())}))
    print({k: (v["launches"], round(v["hbm_bytes_per_launch"] / 1e9, 3), round(v.get("sq_insts_valu_per_launch", 0) / 1e9, 3)) for k, v in kernels.items()
           if v["hbm_bytes_per_launch"] > 1e8})


if __name__ == "__main__":
    main(sys.argv[1])
