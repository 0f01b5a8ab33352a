#!/bin/bash
# rocprofv3 evidence for the general kernel (lights, other primitive kinds, halo) (runs on the GPU box).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/profile_cmd.py r02g_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py r02g_mixed --passes valu,misc -- python3 tools/perf_scene.py --scene mixed --w 1920 --h 1080 --spp 100 --reps 3 &&
python3 tools/profile_cmd.py r02g_spheres_lit --passes valu,misc,mfma -- python3 tools/perf_scene.py --scene random_spheres_lit --w 1200 --h 675 --spp 100 --lights --reps 3
