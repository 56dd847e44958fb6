#!/bin/bash
# tools/profile.sh [tag] [bench.py args...] -- rocprofv3 evidence for bench.py (run on the GPU box via gpurun).
#   pass 1: --kernel-trace --stats           -> per-kernel durations
#   pass 2..: --pmc <counters> (own runs, kernel-trace only, never with sys/hip/hsa traces)
# Results land under gpurun_out/prof_<tag>/; tools/summarize_prof.py copies the summaries worth keeping
# to profiles/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}; shift
P=$R/gpurun_out/prof_$TAG
mkdir -p "$P"
# which binary these counters are read from: the digest of the library's sources (graphik_amd/build.py), carried into
# profiles/<tag>_hbm_traffic.json by summarize_prof.py and compared by bench.py (roofline.traffic_stale)
cp "$R/graphik_amd/lib/libgraphik_amd.so.digest" "$P/source_digest.txt" 2>/dev/null || echo unknown > "$P/source_digest.txt"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --serving-streams 0 --headline-only $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o r1 -- python $R/bench.py $ARGS > $P/kt_bench.json 2> $P/kt.err
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_LDS"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pmc_$n -o r1 -- python $R/bench.py $ARGS > /dev/null 2> $P/pmc_$n.err
  echo "== $TAG $n: $(ls $P/pmc_$n 2>/dev/null | tr '\n' ' ')"; tail -n 2 $P/pmc_$n.err | cut -c1-300
done
