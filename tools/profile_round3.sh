#!/bin/bash
# rocprofv3 evidence of the kernels as they stand in round 3 (runs on the GPU box); the summaries go to profiles/r03/.
# Each tag: one --kernel-trace --stats run plus separate --pmc passes of the same command (tools/profile_cmd.py).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-secondary"
python3 tools/profile_cmd.py r03_c4 --passes valu,misc,mfma,fetch,write -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r03_c3 --passes valu,misc,mfma,fetch,write -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r03_c2 --passes valu,misc,fetch,write -- $B --workload c2 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r03_c5 --passes valu,misc,mfma,fetch,write -- $B --workload c5 --steps 1 --warmup 0 &&
python3 tools/profile_cmd.py r03_c4_nolists --passes valu,mfma --env RT_AMD_PRIM_MIN=0 -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r03_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py r03_mixed --passes valu,misc -- python3 tools/perf_scene.py --scene mixed --w 1920 --h 1080 --spp 100 --reps 3 &&
python3 tools/profile_cmd.py r03_spheres_lit --passes valu,misc,mfma -- python3 tools/perf_scene.py --scene random_spheres_lit --w 1200 --h 675 --spp 100 --lights --reps 3
