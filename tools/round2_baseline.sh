#!/bin/bash
# Baseline evidence: the tree as checked out profiled workload by workload (runs on the GPU box); tags base_*.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 tools/valu_mix.bin > gpurun_out/valu_mix.txt 2>&1 &&
python3 tools/profile_cmd.py base_c3 -- python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline &&
python3 tools/profile_cmd.py base_c2 -- python3 bench.py --workload c2 --steps 5 --warmup 1 --no-cpu-baseline &&
python3 tools/profile_cmd.py base_c3_lds --passes valu,lds --env RT_AMD_SPHERE_SOURCE=lds -- python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline &&
python3 tools/profile_cmd.py base_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py base_c5 --passes valu,misc -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline &&
python3 tools/profile_cmd.py base_c5_lds --passes valu,lds --env RT_AMD_SPHERE_SOURCE=lds -- python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu-baseline
