#!/bin/bash
run() { echo -n "$1 tile=$2: "; RT_AMD_TILE=$2 RT_AMD_LIB=$1 timeout -k 10 100 python -u bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; }
for t in 2 3 4; do run $1 $t; done
for t in 2 3 4; do echo -n "tile=$t "; RT_AMD_TILE=$t timeout -k 10 100 python tools/dbg_stats.py random_spheres 1200 675 500 50 | grep waves; done
