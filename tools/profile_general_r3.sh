#!/bin/bash
# rocprofv3 evidence of the general kernel after the halo pass was cut per tile and the box / polygon tests were trimmed
# (runs on the GPU box); summaries go to profiles/r03/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/profile_cmd.py r03b_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py r03b_mixed --passes valu,misc -- python3 tools/perf_scene.py --scene mixed --w 1920 --h 1080 --spp 100 --reps 3 &&
python3 tools/profile_cmd.py r03b_mixed_selected --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_selected --w 1920 --h 1080 --spp 100 --reps 3
