#!/bin/bash
run() { echo -n "$1 wgs=$2: "; RT_AMD_LIB=$1 RT_AMD_WGS_PER_CU=$2 timeout -k 10 100 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; }
for lib in "$@"; do for w in 5 8; do run $lib $w; done; done
