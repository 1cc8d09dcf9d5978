#!/bin/bash
# On the GPU box: tools/ubench at 4 and 8 resident waves per SIMD, plain (wall time) and under rocprofv3 --pmc, so that every
# probe gets TRUE shader cycles per instruction and the clock it ran at (tooling).  -> gpurun_out/<tag>_ubench.json
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
[ -x $R/tools/ubench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/ubench $R/tools/ubench.hip
for W in ${WAVES:-4 8}; do
  rm -rf $R/gpurun_out/ubench_pmc_w$W
  timeout 300 $R/tools/ubench --waves $W --json > $R/gpurun_out/ubench_w$W.jsonl 2> $R/gpurun_out/ubench_w$W.err; echo "ubench w$W rc=$?"
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY \
      --output-format csv -d $R/gpurun_out/ubench_pmc_w$W -o ub -- $R/tools/ubench --waves $W --json > /dev/null 2> $R/gpurun_out/ubench_pmc_w$W.err; echo "pmc w$W rc=$?"
done
timeout 120 $R/tools/ubench --waves 8 > $R/gpurun_out/${TAG}_ubench.txt 2>&1
cd $R && python tools/ubench_summarize.py $TAG
