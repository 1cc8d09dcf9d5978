#!/usr/bin/env python3
"""tools/ubench_pmc.sh's output -> gpurun_out/<tag>_ubench.json (tooling; runs on the GPU box).

Per probe and per occupancy (4 / 8 waves per SIMD): the wall time of the un-profiled run, and from the rocprofv3 --pmc run
  cycles            GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
  clock_ghz         cycles / kernel duration
  cyc_per_inst      cycles * 1024 SIMDs / SQ_INSTS_VALU            (true shader cycles per wave64 instruction per SIMD)
  valu_busy         SQ_ACTIVE_INST_VALU * 4 / (1024 * cycles)       (quad-cycles the VALU worked / SIMD cycles)
Instruction classes (what bench.py prices a kernel's instruction histogram with) are the medians of their members."""
import csv
import glob
import json
import os
import re
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
SIMDS = 1024

CLASSES = {
    "cheap32": ["v_add_u32", "v_sub_u32", "v_and_b32", "v_xor_b32", "v_lshrrev_b32"],
    "mov": ["v_mov_b32"],
    "mad64": ["v_mad_u64_u32", "v_mad_u64_u32 sgpr", "v_mad_u64_u32 inl", "v_mad_u64_u32 sdst", "v_mad_i64_i32"],
    "carry_vop3": ["v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_add_co_u32_e64", "v_addc_co_u32_e64", "v_lshl_add_u64",
                   "v_lshlrev_b64", "v_lshrrev_b64", "v_cmp_lt_u64", "v_alignbit_b32", "v_perm_b32", "v_bfe_u32", "v_lshl_or_b32", "v_and_or_b32",
                   "v_add3_u32", "v_lshl_add_u32", "v_lshlrev_b32", "v_mul_lo_u32", "v_mul_hi_u32"],
    "cndmask": ["v_cndmask_b32"],
}


def load(w):
    names = {}
    wall = {}
    f = os.path.join(G, "ubench_w%d.jsonl" % w)
    if not os.path.exists(f):
        return None
    for line in open(f):
        line = line.strip()
        if line.startswith("{"):
            d = json.loads(line)
            names[d["op"]] = d["name"]
            wall[d["name"]] = d
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(dict)
    for c in glob.glob(os.path.join(G, "ubench_pmc_w%d" % w, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(c)):
            m = re.search(r"\bk[xy]?<(\d+)>", r["Kernel_Name"])
            if not m:
                continue
            name = names.get(int(m.group(1)))
            if name is None:
                continue
            acc[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            dur[name][int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out = {}
    for name, d in wall.items():
        e = {"ms_unprofiled": d["ms"], "wave_insts_per_simd": d["wave_insts_per_simd"]}
        if name in acc:
            ids = sorted(dur[name])[1:] or sorted(dur[name])  # the first launch of a probe is the warm-up
            def mean(cn):
                per = defaultdict(float)
                for i, v in acc[name].get(cn, []):
                    per[i] += v
                vals = [per[i] for i in ids if i in per]
                return sum(vals) / len(vals) if vals else 0.0
            cycles = mean("GRBM_GUI_ACTIVE") / 8
            ms = sum(dur[name][i] for i in ids) / len(ids)
            insts = mean("SQ_INSTS_VALU")
            e.update({"ms_under_pmc": ms, "cycles": cycles, "clock_ghz": cycles / (ms * 1e6) if ms else 0.0,
                      "sq_insts_valu": insts, "cyc_per_inst": cycles * SIMDS / insts if insts else 0.0,
                      "valu_busy": mean("SQ_ACTIVE_INST_VALU") * 4 / (SIMDS * cycles) if cycles else 0.0,
                      "wait_inst_any_over_wave_cycles": mean("SQ_WAIT_INST_ANY") / mean("SQ_WAVE_CYCLES") if mean("SQ_WAVE_CYCLES") else 0.0})
        out[name] = e
    return out


def main(tag):
    res = {"method": __doc__.split("\n\n")[1], "simds": SIMDS, "occupancy": {}}
    for w in (1, 2, 4, 8):
        probes = load(w)
        if probes is None:
            continue
        classes = {}
        for cname, members in CLASSES.items():
            v = [probes[m]["cyc_per_inst"] for m in members if m in probes and probes[m].get("cyc_per_inst")]
            c = [probes[m]["clock_ghz"] for m in members if m in probes and probes[m].get("clock_ghz")]
            if v:
                classes[cname] = {"cyc_per_inst_median": statistics.median(v), "min": min(v), "max": max(v), "clock_ghz_median": statistics.median(c), "n": len(v)}
        res["occupancy"]["w%d" % w] = {"probes": probes, "classes": classes}
    json.dump(res, open(os.path.join(G, "%s_ubench.json" % tag), "w"), indent=1)
    for w, o in res["occupancy"].items():
        print(w, {k: (round(v["cyc_per_inst_median"], 2), round(v["min"], 2), round(v["max"], 2), round(v["clock_ghz_median"], 2)) for k, v in o["classes"].items()})
        for n in ("mix hash_leaves", "mix limb_ntt", "v_mad_u64_u32", "v_add_u32"):
            p = o["probes"].get(n, {})
            print("  ", n, {k: round(p[k], 3) for k in ("ms_unprofiled", "ms_under_pmc", "clock_ghz", "cyc_per_inst", "valu_busy") if k in p})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
