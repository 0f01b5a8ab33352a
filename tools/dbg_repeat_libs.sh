#!/bin/bash
# usage: dbg_repeat_libs.sh REPS lib1.so lib2.so ...  -- tools/dbg_diff_tiles.py (lit field of 5000 spheres, 960x540, 64 spp) with each library
reps=$1; shift
for l in "$@"; do echo "== $l"; RT_AMD_LIB=$PWD/$l timeout -k 10 200 python -u tools/dbg_diff_tiles.py 5000 960 540 64 $reps 2>&1 | grep "^run" | awk '{n++; if ($NF != 0) bad++} END {print n " frames, " bad+0 " differ"}' || exit 1; done
