#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...>   -- one rocprofv3 --pmc pass over a 1-step bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(float); n=0
for f in glob.glob('gpurun_out/pmc_$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in sorted(agg.items()): print(f"{k} {v:.5g}")
PY
