#!/bin/bash
# The rest of the round's last profiles (lists off, the general kernel's scenes), with a c4 bench line first to place the box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-secondary"
timeout -k 10 200 $B --workload c4 > gpurun_out/bench_c4_b.json 2>/dev/null || exit 1
python3 tools/profile_cmd.py r03_c4_nolists --passes valu,mfma --env RT_AMD_PRIM_MIN=0 -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r03_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py r03_mixed --passes valu,misc -- python3 tools/perf_scene.py --scene mixed --w 1920 --h 1080 --spp 100 --reps 3 &&
python3 tools/profile_cmd.py r03_mixed_selected --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_selected --w 1920 --h 1080 --spp 100 --reps 3 &&
python3 tools/profile_cmd.py r03_spheres_lit --passes valu,misc,mfma -- python3 tools/perf_scene.py --scene random_spheres_lit --w 1200 --h 675 --spp 100 --lights --reps 3
