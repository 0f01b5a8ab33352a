#!/bin/bash
# Kernel time of the benchmark workloads under every work-item tile shape (RT_AMD_TILE 0..6), one library.
# usage: tools/tile_sweep.sh <lib.so> "<workloads>" "<tiles>"
lib=$1; wls=${2:-"c2 c3 c4 c5"}; tiles=${3:-"0 1 2 3 4 5 6"}
for wl in $wls; do
  steps=5; [ $wl = c4 ] && steps=2; [ $wl = c5 ] && steps=1
  for t in $tiles; do
    echo -n "$wl tile=$t "
    RT_AMD_LIB=$PWD/$lib RT_AMD_TILE=$t timeout -k 10 200 python3 bench.py --workload $wl --steps $steps --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
  done
done
