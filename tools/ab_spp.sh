#!/bin/bash
# usage: ab_spp.sh libA libB  -- c3 at several sample counts with each library
run() { echo -n "$1 spp=$2: "; RT_AMD_LIB=$1 timeout -k 10 100 python -u bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --spp $2 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; }
for spp in 250 500 1000 2000; do for lib in "$@"; do run $lib $spp; done; done
