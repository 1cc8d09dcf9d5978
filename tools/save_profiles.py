#!/usr/bin/env python3
"""Copy the rocprofv3 / PMC summaries of the last `tools/gpu_run.sh prof pmc` + `tools/pmc_sq.sh` run from
gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/save_profiles.py <tag> [sq_insts_valu_hash]"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1]
with open(os.path.join(G, "prof", "p2hot_kernel_stats.csv")) as f:
    lines = f.readlines()
with open(os.path.join(P, "%s_kernel_stats.csv" % tag), "w") as f:
    f.writelines([l for l in lines[:1]] + [l for l in lines[1:] if "at::native" not in l][:8])
d = json.load(open(os.path.join(G, "pmc_summary.json")))
sq = {}
sqf = os.path.join(G, "pmc_sq_summary.txt")
if os.path.exists(sqf):
    shutil.copy(sqf, os.path.join(P, "%s_pmc_sq.txt" % tag))
    for line in open(sqf):
        m = re.match(r"(\w+) launches (\d+) ms ([\d.]+) (\{.*\})", line)
        if m:
            sq[m.group(1)] = (int(m.group(2)), float(m.group(3)), eval(m.group(4)))
out = {"workload": {"W": 135, "log_n": 20, "rate_bits": 3, "cap_height": 4, "n_gpus": 1},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `bench.py --steps 1 --warmup 0`; "
                 "FETCH_SIZE doubled (gfx950 tallies 128-B reads at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as "
                 "reported; KiB -> bytes.  SQ_INSTS_VALU / GRBM_GUI_ACTIVE from a third pass (tools/pmc_sq.sh).",
       "kernels": d}
if "hash_leaves" in sq:
    n, ms, c = sq["hash_leaves"]
    out["kernels"]["hash_leaves_kernel"]["sq_insts_valu_per_launch"] = float(c["SQ_INSTS_VALU"]) / n
    out["kernels"]["hash_leaves_kernel"]["clock_ghz"] = float(c["GRBM_GUI_ACTIVE"]) / 8 / (ms * 1e6)
json.dump(out, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(P, "%s_pmc_traffic.json" % tag), "w"), indent=1)
if os.path.exists(os.path.join(G, "bench.json")):
    shutil.copy(os.path.join(G, "bench.json"), os.path.join(P, "%s_bench.json" % tag))
print({k: (v["launches"], round(v["hbm_bytes_per_launch"] / 1e9, 3)) for k, v in d.items()})
