#!/bin/bash
# usage: ab_wl.sh libA libB  -- c3 and c4 with each library in one call, with clocks sampled mid-run for c4
run() { echo -n "$1 $2: "; RT_AMD_LIB=$1 timeout -k 10 100 python -u bench.py --workload $2 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; }
for wl in c3 c4 c3 c2; do for lib in "$@"; do run $lib $wl; done; done
rocm-smi --showpower --showtemp 2>/dev/null | grep -E "Power|Temp" | head -8
