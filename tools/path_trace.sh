#!/bin/bash
# On the GPU box: rocprofv3 kernel trace of tools/path_probe.py (the non-commit stages of the per-proof path), per-kernel averages (tooling).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_prof -o pp -- python $R/tools/path_probe.py ${1:-20} > /tmp/pp_probe.log 2>&1
f=$(find /tmp/pp_prof -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { tail -20 /tmp/pp_probe.log; exit 1; }
python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("pp_", "eval_polys", "ext_powers", "reduce_polys", "horner", "divide", "quot", "fold", "shift_acc")):
        print("%-72s calls %5s avg %10.1f us total %8.3f ms" % (n[:72], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
