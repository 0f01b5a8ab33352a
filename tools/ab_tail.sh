#!/bin/bash
run() { echo -n "$1 tail=$2: "; RT_AMD_TAIL_PCT=$2 RT_AMD_LIB=$1 timeout -k 10 100 python -u bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; }
run tools/v_new2.so 0
for t in 0 10 25 40; do run $1 $t; done
for t in 0 25; do echo -n "tail=$t "; RT_AMD_TAIL_PCT=$t timeout -k 10 100 python tools/dbg_stats.py random_spheres 1200 675 500 50 | grep waves; done
