#!/bin/bash
# rocprofv3 evidence for the matrix-core filter (runs on the GPU box): kernel-trace stats + PMC passes for the
# sphere-only workloads, with the matrix-pipe counters.  Summaries land in gpurun_out/<tag>/summary.json.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --no-cpu-baseline --no-secondary"
python3 tools/profile_cmd.py r02m_c4 --passes valu,misc,mfma,fetch,write -- $B --workload c4 --steps 3 --warmup 1 &&
python3 tools/profile_cmd.py r02m_c3 --passes valu,misc,mfma,fetch,write -- $B --workload c3 --steps 5 --warmup 1 &&
python3 tools/profile_cmd.py r02m_c5 --passes valu,misc,mfma -- $B --workload c5 --steps 1 --warmup 0
