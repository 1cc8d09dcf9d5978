cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_sq -o p2hot -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_sq.err
echo rc=$?
python - <<'P'
import csv,glob,collections,os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob(R+'/gpurun_out/pmc_sq/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        for key in ('hash_leaves','merkle_level','regpass','bitrev'):
            if key in k:
                acc[key][r['Counter_Name']]+=float(r['Counter_Value']); 
                if r['Counter_Name']=='GRBM_GUI_ACTIVE': cnt[key]+=1
                acc[key]['_dur_ns']+= (int(r['End_Timestamp'])-int(r['Start_Timestamp'])) if r['Counter_Name']=='GRBM_GUI_ACTIVE' else 0
for k,v in acc.items():
    print(k,cnt[k],{a:("%.4g"%b) for a,b in v.items()})
    if v.get('_dur_ns'): print("   clock GHz ~", v['GRBM_GUI_ACTIVE']/v['_dur_ns'])
P
