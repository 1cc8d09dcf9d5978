#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/sq_counters.txt
for v in 0 1 2 3 4 7; do
  P2HOT_LIMB_DEBUG=$v timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/dbg_$v.json 2> gpurun_out/dbg_$v.err
  python - "$v" gpurun_out/dbg_$v.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("debug", sys.argv[1], {k:round(x["ms_per_launch"],3) for k,x in d["kernels"].items() if "ntt" in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open(sys.argv[2].replace("json","err")).read()[-300:])
PY
done
