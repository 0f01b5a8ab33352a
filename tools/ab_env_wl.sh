#!/bin/bash
# usage: ab_env_wl.sh WORKLOAD STEPS VAR v1 v2 ...   -- one bench line per value of an environment knob
wl=$1; steps=$2; var=$3; shift 3
for v in "$@"; do echo -n "$var=$v: "; env $var=$v timeout -k 10 200 python -u bench.py --workload $wl --steps $steps --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms, scanned share',d['roofline'].get('scanned_ray_share'))" || exit 1; done
