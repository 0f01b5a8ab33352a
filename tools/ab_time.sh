#!/bin/bash
# Times a workload with several builds of the library on one box, interleaved (boxes differ by 1-2 %).
# usage: tools/ab_time.sh <workload> <libA.so> <libB.so> ...
set -e
wl=$1; shift
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload $wl --steps ${STEPS:-5} --warmup ${WARMUP:-1} --no-secondary --no-cpu-baseline 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', d['ms_per_step'], d['config'].get('rays_per_sample'))"
  done
done
