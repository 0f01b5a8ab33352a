#!/bin/bash
# rocprofv3 evidence of the kernels as they stand (runs on the GPU box); the summaries go to profiles/r02/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-secondary"
python3 tools/profile_cmd.py r02f_c4 --passes valu,misc,mfma,fetch,write -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r02f_c3 --passes valu,misc,mfma,fetch,write -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02f_c2 --passes valu,misc,fetch,write -- $B --workload c2 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02f_c5 --passes valu,misc,mfma -- $B --workload c5 --steps 1 --warmup 0 &&
python3 tools/profile_cmd.py r02f_mixed_lit --passes valu,misc -- python3 tools/perf_scene.py --scene mixed_lit --w 1920 --h 1080 --spp 100 --lights --reps 3 &&
python3 tools/profile_cmd.py r02f_mixed --passes valu,misc -- python3 tools/perf_scene.py --scene mixed --w 1920 --h 1080 --spp 100 --reps 3
