#!/bin/bash
# usage: ab_c5.sh lib1.so lib2.so ...  -- c5 once per library: frame time, scans, and time per table block per wave
for lib in "$@"; do echo -n "$lib: "; RT_AMD_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric"' | python -c "
import sys,json,re
d=json.loads(sys.stdin.read()); r=d['roofline']; m=re.search(r'(\d+) table rows.*scans/launch=(\d+)', r['flop_model'])
rows,scans=int(m.group(1)),int(m.group(2)); blocks=scans*rows/32/4096
print(d['ms_per_step'],'ms; scans',scans,'; rays/sample',d['config']['rays_per_sample'],'; ns per block per wave %.1f' % (r['kernel_ms']*1e6/blocks))" || exit 1; done
