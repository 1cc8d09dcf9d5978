#!/bin/bash
# Runs bench.py as N processes that SHARE the box's one GPU (gloo process group, host-staged exchange): exercises the
# whole multi-process flow of the scaling runs -- launcher env, column ranges, sharded commit, cap all-gather, the cap check
# against the golden of the N-GPU shape -- where no N-GPU node is at hand.  Not a scaling measurement (tooling).
# usage: tools/gpu_multiproc.sh N [log_n per rank, default 20] [extra bench args]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-2}; LOGN=${2:-20}; shift; shift
mkdir -p gpurun_out
export P2HOT_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
    bench.py --gpus "$N" --steps 1 --warmup 1 --log-n "$LOGN" "$@" > "gpurun_out/multiproc_$N.json" 2> "gpurun_out/multiproc_$N.err"
echo "multiproc N=$N rc=$?"
python - "$N" <<'P'
import json, sys
try:
    d = json.loads(open("gpurun_out/multiproc_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print({k: d[k] for k in ("n_gpus", "ms_per_step", "cap_checked")}, d["config"]["workload"][:60], "|", d["config"]["transport"][:40])
except Exception as e:
    print("no JSON line:", e)
P
tail -5 "gpurun_out/multiproc_$N.err"
