#!/bin/bash
# bench.py as N processes that SHARE the box's one GPU (gloo process group, host-staged exchange): the whole multi-process flow of
# the scaling runs where no N-GPU node is at hand.  Not a scaling measurement (tooling).  bench.py launches its own ranks.
# usage: tools/gpu_multiproc.sh N [log_n per rank, default 20] [extra bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out
N=${1:-2}; LOGN=${2:-20}; shift; shift
P2HOT_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus "$N" --steps 1 --warmup 1 --log-n "$LOGN" "$@" \
    > "gpurun_out/multiproc_$N.json" 2> "gpurun_out/multiproc_$N.err"; echo "multiproc N=$N rc=$?"; tail -c 600 "gpurun_out/multiproc_$N.json"
