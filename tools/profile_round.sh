#!/bin/bash
# Collects the rocprofv3 evidence for the default bench command (runs on the GPU box).
# usage: tools/profile_round.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/stats.log 2>&1 || exit 1
pass() { name=$1; shift; timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$name.log 2>&1 || exit 1; }
pass valu SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass misc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 - <<PY
import csv, glob, collections, json
res={}
for name in ('valu','misc','fetch','write'):
    agg=collections.defaultdict(float)
    for f in glob.glob('$out/pmc_%s/*/*counter_collection.csv'%name):
        for r in csv.DictReader(open(f)):
            if 'render_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']]+=float(r['Counter_Value'])
    res[name]=dict(agg)
json.dump(res, open('$out/pmc_summary.json','w'), indent=1)
print(json.dumps(res, indent=1))
PY
cat $out/stats/*/*kernel_stats.csv | head -3
tail -1 $out/stats.log | cut -c1-900
