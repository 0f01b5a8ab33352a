#!/bin/bash
# usage: ab_env.sh VAR v1 v2 ...   -- c3 bench with an environment knob at several values (twice each, interleaved)
var=$1; shift
for rep in 1 2; do for v in "$@"; do echo -n "$var=$v: "; env $var=$v timeout -k 10 100 python -u bench.py --workload c3 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"; done; done
