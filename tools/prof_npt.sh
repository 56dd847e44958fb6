#!/bin/bash
# tools/prof_npt.sh [B] -- PMC counters of the table-scene solve kernels at B resident problems (default 64:
# one problem per CU, i.e. the latency of a lone problem): node-per-lane kernel (two wavefronts / one
# wavefront per problem) and the workgroup kernel, through tools/npt_check.py.  gpurun_out/prof_npt/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/prof_npt
mkdir -p "$P"
export NPT_B=${1:-64}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_F64"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pmc_$n -o r1 -- python $R/tools/npt_check.py timing > $P/$n.out 2> $P/pmc_$n.err
  tail -n 1 $P/pmc_$n.err | cut -c1-200
done
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$P/pmc_*/r1_counter_collection.csv")):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        acc[(row["Dispatch_Id"], row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])] += float(row["Counter_Value"])
    for (d, k, c), v in acc.items():
        if "rtr_" in k: per[k][c].append(v)
for k, cs in per.items():
    print(k)
    for c, v in sorted(cs.items()): print("   %-24s last launch %.4g   (launches %d)" % (c, v[-1], len(v)))
PY
grep path $P/SQ_INSTS_VALU.out
