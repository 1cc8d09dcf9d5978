#!/bin/bash
# On the GPU box: A/B of two builds of libp2hot on the headline step (tooling).  The candidate is the in-tree library, the
# baseline a second build shipped as tools/ab/<name>.so.  Driver-style timings alternate A B A B; then one PMC pass each.
# usage: tools/ab_lib.sh <tag> <baseline .so under tools/ab> "<pmc counters>"
TAG=$1; BASE=$2; CNT="${3:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS}"
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
cp plonky2_amd/libp2hot.so /tmp/cand.so
use() { cp "$1" plonky2_amd/libp2hot.so; touch plonky2_amd/libp2hot.so; }
line() { python - "$1" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step %.3f" % d["ms_per_step"], "cap_checked", d["cap_checked"], {k: round(v["ms_per_launch"], 3) for k, v in d["kernels"].items() if k.startswith("ntt_") or k.startswith("bitrev")})
P
}
for rep in 1 2; do
  for v in cand base; do
    if [ $v = cand ]; then use /tmp/cand.so; else use tools/ab/$BASE; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/${TAG}_${v}_${rep}.json 2> gpurun_out/${TAG}_${v}_${rep}.err
    line gpurun_out/${TAG}_${v}_${rep}.json
  done
done
cd /tmp && export TMPDIR=/tmp
for v in cand base; do
  if [ $v = cand ]; then (cd $R && use /tmp/cand.so); else (cd $R && use tools/ab/$BASE); fi
  rm -rf $R/gpurun_out/pmc_$v
  timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$v -o p2hot -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $R/gpurun_out/pmc_$v.err
  echo "pmc $v rc=$?"
  python - $R/gpurun_out/pmc_$v <<'P' | tee $R/gpurun_out/${TAG}_pmc_$v.txt
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.Counter(); seen = set()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        for key in ('hash_leaves', 'merkle_level', 'limbpass', 'bitrev'):
            if key in k:
                if key == 'limbpass': key += k[k.index('<'):k.index('>') + 1].replace(' ', '')
                acc[key][r['Counter_Name']] += float(r['Counter_Value'])
                if r['Dispatch_Id'] not in seen:
                    seen.add(r['Dispatch_Id']); cnt[key] += 1; dur[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in sorted(acc.items()):
    print(k, 'launches', cnt[k], 'ms', round(dur[k] / 1e6, 3), {a: ("%.4g" % b) for a, b in v.items()})
P
done
cd $R && use /tmp/cand.so
