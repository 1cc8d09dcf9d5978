#!/bin/bash
# A/B on the GPU box: NTT parity tests, then the bench under each listed environment (tooling)
# usage: tools/gpu_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument is one variant's environment)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity.py -m gpu -x -q -k "fft or lde or ntt or coset or commit" 2>&1 | tail -4
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  python - "$v" gpurun_out/ab_$i.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print(sys.argv[1], "| GFE/s %.3f ms %.3f cap %s" % (d["value"], d["ms_per_step"], d.get("cap_checked")), {k:round(x["ms_per_launch"],3) for k,x in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  grep -v amdgpu.ids gpurun_out/ab_$i.err | tail -2
done
