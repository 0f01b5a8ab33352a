#!/bin/bash
# A/B timing of library builds in one GPU call: ab.sh libA libB ...  (each run twice, interleaved)
run() { echo -n "$1: "; RT_AMD_LIB=$1 timeout -k 10 100 python -u bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; }
for rep in 1 2; do for lib in "$@"; do run $lib; done; done
