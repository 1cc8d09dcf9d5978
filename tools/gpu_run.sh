#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, instruction-rate probe, bench, rocprof.
# usage: tools/gpu_run.sh [tests] [ubench] [bench] [prof] [pmc]   (default: all but pmc)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${*:-tests ubench bench prof}"
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -40 > gpurun_out/pytest_gpu.log; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log ;;
ubench)
  timeout 120 ./tools/ubench > gpurun_out/ubench.log 2>&1; cat gpurun_out/ubench.log ;;
bench)
  timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
  lscpu | egrep "Model name|^CPU\(s\)|Socket|Thread" > gpurun_out/lscpu.txt ;;
prof)
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o p2hot -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err" ); echo "prof rc=$?"
  find gpurun_out/prof -name "*kernel_stats*" | head -3
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 ;;
pmc)
  ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_fetch" -o p2hot -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OLDPWD/gpurun_out/pmc1.err" ); echo "pmc fetch rc=$?"
  ( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OLDPWD/gpurun_out/pmc_write" -o p2hot -- python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OLDPWD/gpurun_out/pmc2.err" ); echo "pmc write rc=$?"
  find gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head; f=$(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && head -5 "$f"
  python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.json 2> gpurun_out/pmc_summary.err; cat gpurun_out/pmc_summary.json; tail -3 gpurun_out/pmc_summary.err ;;
esac
done
