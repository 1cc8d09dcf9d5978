#!/bin/bash
# On the GPU box: the asynchronous leaf copy of the C3 wires commit (other_configs.host_c3_wires_leaves_async) under the pipeline
# knobs -- lanes split or not, leaf blocks issued before the digests' copy (tooling).   usage: tools/async_sweep.sh "1:0 1:4 1:8 0:0 0:16"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in ${1:-1:0 1:4 1:16 0:0 0:16}; do
  split=${cfg%%:*}; early=${cfg##*:}
  P2HOT_HOST_ASYNC_SPLIT=$split P2HOT_HOST_ASYNC_EARLY_BLOCKS=$early python bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-only host 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); oc=d['other_configs']; a=oc['host_c3_wires_leaves_async']
print('split=$split early_blocks=$early: call %.1f ms, first block %.1f, last row %.1f | sync pinned %.1f, no leaves %.1f, cap %s' % (a['ms'], a['ms_first_block'], a['ms_last_row'], oc['host_c3_wires_leaves_back_pinned']['ms'], oc['host_c3_wires_coeffs_digests']['ms'], a['cap_checked']))"
done
