#!/bin/bash
# usage: ab_libs.sh WORKLOAD STEPS lib1.so lib2.so ...   -- one bench line per library (RT_AMD_LIB), twice, interleaved
wl=$1; steps=$2; shift 2
for rep in 1 2; do for lib in "$@"; do echo -n "$lib: "; RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --workload $wl --steps $steps --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms')" || exit 1; done; done
