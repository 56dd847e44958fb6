#!/bin/bash
# dev: PMC counters of the workgroup-per-problem path on the UR10 + table scene
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/prof_table
mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_F64"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 250 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pmc_$n -o r1 -- python $R/tools/attic/dev_table_seg.py > $P/$n.out 2> $P/pmc_$n.err
  tail -n 1 $P/pmc_$n.err | cut -c1-200
done
