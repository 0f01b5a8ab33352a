#!/bin/bash
# usage: tools/gpu_check.sh [pytest|bench|both]   (runs on the GPU box)
set -e
mode=${1:-both}
if [ "$mode" != "bench" ]; then
  timeout -k 10 400 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread 2>&1 | tail -15
fi
if [ "$mode" != "pytest" ]; then
  timeout -k 10 200 python -u bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SCALAR', d['value'],'Ms/s', d['ms_per_step'],'ms frac',d['roofline']['frac'], 'rays/sample', d['config']['rays_per_sample'])"
  RT_AMD_SPHERE_SOURCE=lds timeout -k 10 200 python -u bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LDS   ', d['value'],'Ms/s', d['ms_per_step'],'ms frac',d['roofline']['frac'])"
fi
