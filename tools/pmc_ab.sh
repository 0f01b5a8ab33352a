#!/bin/bash
# usage: tools/pmc_ab.sh libA libB ...   -- one VALU-counter pass per library over a 1-step bench run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  tag=$(basename $lib .so)
  RT_AMD_LIB=$lib timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(float)
for f in glob.glob('gpurun_out/pmc_$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
print('$tag', ' '.join(f"{k}={v:.4g}" for k,v in sorted(agg.items())))
PY
done
