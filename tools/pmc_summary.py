#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Units: both counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced streaming read (TCC_EA0_RDREQ tallied at 64 B per 128-B request),
so the read side is doubled -- WRITE_SIZE is used as is (uncalibrated, see the guide).
usage: pmc_summary.py <fetch_dir> <write_dir>   -> JSON {kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch, ...}}"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r.get("Kernel_Name", "?")
            acc[k][0] += float(r.get("Counter_Value", 0))
            acc[k][1] += 1
    return acc


def short(name):
    for key in ("hash_leaves_kernel", "merkle_level_kernel", "ntt_regpass_kernel", "ntt_pass_kernel", "bitrev_permute_kernel", "transpose_kernel"):
        if key in name:
            return key
    return None


def main(fd, wd):
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    out = {}
    for name in set(fe) | set(wr):
        s = short(name)
        if not s:
            continue
        f, w = fe.get(name, [0, 0]), wr.get(name, [0, 0])
        n = max(f[1], w[1], 1)
        e = out.setdefault(s, {"launches": 0, "fetch_size_kib_raw": 0.0, "write_size_kib_raw": 0.0})
        e["launches"] += n
        e["fetch_size_kib_raw"] += f[0]
        e["write_size_kib_raw"] += w[0]
    for s, e in out.items():
        n = e["launches"]
        e["fetch_bytes_per_launch"] = 2.0 * e["fetch_size_kib_raw"] * 1024 / n  # gfx950 correction (x2)
        e["write_bytes_per_launch"] = e["write_size_kib_raw"] * 1024 / n
        e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
