#!/bin/bash
# Times tools/perf_scene.py on a scene with several builds of the library, interleaved.
# usage: tools/ab_scene.sh "<perf_scene args>" <libA.so> <libB.so> ...
set -e
args=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "; RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python3 tools/perf_scene.py $args 2>/dev/null | tail -1 | cut -c1-70
  done
done
