#!/bin/bash
# The round's last profiles: bench lines and rocprofv3 passes of the same workloads in one call, so that both come from one box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-secondary"
for wl in c4 c3 c2 c5; do timeout -k 10 200 $B --workload $wl > gpurun_out/bench_$wl.json 2>/dev/null || exit 1; done
python3 tools/profile_cmd.py r03_c4 --passes valu,misc,mfma,fetch,write -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r03_c3 --passes valu,misc,mfma,fetch,write -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r03_c2 --passes valu,misc,fetch,write -- $B --workload c2 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r03_c5 --passes valu,misc,mfma,fetch,write -- $B --workload c5 --steps 1 --warmup 0 &&
for wl in c4 c3; do timeout -k 10 200 $B --workload $wl > gpurun_out/bench_${wl}_after.json 2>/dev/null || exit 1; done
