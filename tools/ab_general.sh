#!/bin/bash
# Times the general kernel's scenes with several builds of the library on one box, interleaved.
# usage: tools/ab_general.sh <libA.so> <libB.so> ...
set -e
for round in 1 2; do
  for lib in "$@"; do
    for sc in "mixed_lit --lights" "mixed_selected" "mixed" "halo_overlap" "random_spheres_lit_floor --lights"; do
      echo -n "$lib: "
      RT_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 tools/perf_scene.py --scene $sc --w 1920 --h 1080 --spp 100 --reps 3 | cut -c1-100
    done
  done
done
