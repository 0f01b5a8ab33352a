#!/bin/bash
# rocprofv3 evidence (runs on the GPU box): kernel-trace stats + PMC passes for every workload,
# the LDS-vs-scalar sphere source A/B, and the general kernel.  Summaries land in gpurun_out/<tag>/summary.json.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-secondary"
python3 tools/profile_cmd.py r02_c4 -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r02_c3 -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02_c2 -- $B --workload c2 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02_c5 --passes valu,misc -- $B --workload c5 --steps 1 --warmup 0 &&
python3 tools/profile_cmd.py r02_c3_lds --passes valu,lds --env RT_AMD_SPHERE_SOURCE=lds -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02_c3_scalar_ldscounters --passes lds -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02_c5_lds --passes valu,lds --env RT_AMD_SPHERE_SOURCE=lds -- $B --workload c5 --steps 1 --warmup 0 &&
python3 tools/profile_cmd.py r02_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py r02_mixed --passes valu,misc -- python3 tools/perf_scene.py --scene mixed --w 1920 --h 1080 --spp 100 --reps 3 &&
python3 tools/profile_cmd.py r02_first_hit --kernel first_hit_kernel --passes valu -- python3 tools/perf_first_hit.py
