#!/bin/bash
# Compile-time A/B/... on the GPU box (tooling): NTT parity tests on the main library, then the bench with each
# plonky2_amd/libp2hot_<tag>.so swapped in for plonky2_amd/libp2hot.so, interleaved ROUNDS times so that the box's
# thermal drift hits every variant alike.
# usage: tools/gpu_abn.sh [ROUNDS] ; variants = main + every plonky2_amd/libp2hot_*.so present
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
ROUNDS=${1:-2}
timeout 900 python -m pytest tests/test_parity.py -m gpu -x -q -k "fft or lde or ntt or coset or commit" 2>&1 | tail -3
cp plonky2_amd/libp2hot.so /tmp/libp2hot_main.so
for r in $(seq 1 $ROUNDS); do
  for lib in /tmp/libp2hot_main.so plonky2_amd/libp2hot_*.so; do
    tag=$(basename $lib .so)
    cp $lib plonky2_amd/libp2hot.so
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/abn_${tag}_$r.json 2> gpurun_out/abn_${tag}_$r.err
    python - "$tag r$r" gpurun_out/abn_${tag}_$r.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print(sys.argv[1], "| GFE/s %.3f ms %.3f cap %s" % (d["value"], d["ms_per_step"], d.get("cap_checked")), {k:round(x["ms_per_launch"],3) for k,x in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
    grep -v amdgpu.ids gpurun_out/abn_${tag}_$r.err | tail -2
  done
done
cp /tmp/libp2hot_main.so plonky2_amd/libp2hot.so
