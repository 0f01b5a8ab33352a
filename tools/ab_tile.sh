#!/bin/bash
# usage: ab_tile.sh lib workload tiles...   -- forced work-item tile shapes (0 = 1x1 .. 4 = 32x8)
lib=$1; wl=$2; shift 2
for t in "$@"; do echo -n "$wl tile=$t: "; RT_AMD_TILE=$t RT_AMD_LIB=$lib timeout -k 10 100 python -u bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; done
