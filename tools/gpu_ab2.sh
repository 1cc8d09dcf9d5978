#!/bin/bash
# like gpu_ab.sh, then the same variants with plonky2_amd/libp2hot_alt.so swapped in (compile-time A/B; tooling)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu_ab.sh "$@"
cp plonky2_amd/libp2hot.so /tmp/libp2hot_main.so; cp plonky2_amd/libp2hot_alt.so plonky2_amd/libp2hot.so
echo "--- alt library"
bash tools/gpu_ab.sh "$@"
cp /tmp/libp2hot_main.so plonky2_amd/libp2hot.so
