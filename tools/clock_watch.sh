#!/bin/bash
# usage: tools/clock_watch.sh lib...  -- runs a ~10 s render with each library while sampling rocm-smi clocks and power
for lib in "$@"; do
  echo "== $lib"
  RT_AMD_LIB=$lib timeout -k 10 120 python -u bench.py --workload c4 --steps 6 --warmup 1 --no-cpu-baseline > gpurun_out/cw_$(basename $lib .so).log 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr '\n' ' '; echo
    sleep 1
  done
  wait $pid
  grep '^{"metric"' gpurun_out/cw_$(basename $lib .so).log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'],'ms frac',d['roofline']['frac'])"
done
