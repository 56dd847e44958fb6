#!/bin/bash
# rocprofv3 kernel-trace + PMC summaries of every BASELINE workload (run on the GPU box through gpurun;
# then `python tools/summarize_prof.py <tag>` here for each tag):
#   r06          c2  LWA4D 4096            (the headline)
#   r06_c3       c3  UR10 + table 4096     (node-per-lane solve kernel)
#   r06_c4       c4  KUKA 65536 on one GPU
#   r06_c4share  c4  KUKA 8192 = the per-GPU share of an 8-GPU run
#   r06_c5       c5  planar-10 65536
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $R/tools/profile.sh r06
bash $R/tools/profile.sh r06_c4 --config c4
bash $R/tools/profile.sh r06_c4share --robot kuka --batch 8192
bash $R/tools/profile.sh r06_c5 --config c5
bash $R/tools/profile.sh r06_c3 --config c3
# the column-form product on the two configs it concerns (bench.py: c2_column / c4_column)
bash $R/tools/profile.sh r06_column --hessian-form column
bash $R/tools/profile.sh r06_c4_column --config c4 --hessian-form column
