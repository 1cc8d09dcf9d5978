#!/bin/bash
# Pass-planning sweep for the large transforms (tooling): bits per strided pass x XCD-aware tile order, at 2^k rows.
# usage: tools/tune_strided.sh "20 22 23" "9 10 11"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in ${1:-22 23}; do for b in ${2:-9 10 11}; do for x in 0 1; do
P2HOT_NTT_STRIDED_BITS=$b P2HOT_NTT_XCD_REMAP=$x python bench.py --steps 3 --warmup 1 --log-n $k --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('k=$k bits=$b remap=$x', round(d['ms_per_step'],1),'ms', round(d['value'],2),'GFE/s cap', d['cap_checked'], {k:(round(v['ms_per_launch'],2),v['launches']) for k,v in d['kernels'].items() if 'ntt' in k})
"
done; done; done
