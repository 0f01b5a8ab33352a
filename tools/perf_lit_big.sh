#!/bin/bash
# Tuning aid: the general kernel on lit clouds too long for a resident table (streamed tiles) against the packed filter.
for n in 1400 2000 5000 10000; do
  timeout -k 10 120 python3 tools/perf_lit_sizes.py --child $n || exit 1
  RT_AMD_SPHERE_SOURCE=scalar timeout -k 10 120 python3 tools/perf_lit_sizes.py --child $n || exit 1
done
