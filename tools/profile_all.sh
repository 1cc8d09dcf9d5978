#!/bin/bash
# On the GPU box: the rocprofv3 kernel trace and the three PMC passes of the headline step, summarised (tooling).
# usage: tools/profile_all.sh <tag>    -> gpurun_out/<tag>_{kernel_stats.csv,pmc_traffic.json,pmc_sq.txt,bench.json}
# Counter passes run alone (--pmc without any trace domain), each in its own process, as the guide prescribes.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out && rm -rf $R/gpurun_out/prof $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_sq2 $R/gpurun_out/pmc_sq3
B="python $R/bench.py --no-cpu-baseline --no-extra"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o p2hot -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err; echo "trace rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o p2hot -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc1.err; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o p2hot -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc2.err; echo "write rc=$?"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc_sq -o p2hot -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc3.err; echo "sq rc=$?"
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 --output-format csv -d $R/gpurun_out/pmc_sq2 -o p2hot -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc4.err; echo "sq2 rc=$?"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_sq3 -o p2hot -- $B --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/pmc5.err; echo "sq3 rc=$?"
cd $R && python tools/prof_summarize.py $TAG && timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_ntt'), {k:round(v['ms_per_launch'],3) for k,v in d['kernels'].items()})"
