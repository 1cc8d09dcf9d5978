#!/bin/bash
# usage: tools/pmc_sq.sh "<counter list>"  -> per-kernel sums (tooling)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CNT="${1:-GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU}"
rm -rf $R/gpurun_out/pmc_sq
timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_sq -o p2hot -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> $R/gpurun_out/pmc_sq.err
echo rc=$?
python - <<'P'
import csv,glob,collections,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.Counter(); seen=set()
for f in glob.glob(R+'/gpurun_out/pmc_sq/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('hash_leaves','merkle_level','regpass','limbpass','bitrev'):
            if key in k:
                if key in ('regpass','limbpass'): key += k[k.index('<'):k.index('>')+1].replace(' ','') + '_g%s' % r['Grid_Size']
                acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
                if r['Dispatch_Id'] not in seen:
                    seen.add(r['Dispatch_Id']); cnt[key]+=1; dur[key]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in sorted(acc.items()):
    print(k,'launches',cnt[k],'ms',dur[k]/1e6,{a:("%.4g"%b) for a,b in v.items()})
P
