# HBM traffic of the planar solve kernel (c5) from two PMC passes:  bash tools/attic/c5_traffic.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/c5traffic_$1; mkdir -p $P
for set in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pmc_$set -o r1 -- python $R/bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --serving-streams 0 --headline-only > $P/bench_$set.json 2> $P/err_$set.txt
done
python - <<PY
import csv,glob
for k in ("rtr_quad_kernel","prep_quad_kernel"):
    tot={}
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        v=[float(r["Counter_Value"]) for f in glob.glob("$P/pmc_%s/*counter_collection.csv"%c) for r in csv.DictReader(open(f)) if k in r["Kernel_Name"] and r["Counter_Name"]==c]
        tot[c]=sum(v)/max(len(v),1)
    print("$1", k, "fetch MB %.1f write MB %.1f"%(tot["FETCH_SIZE"]*2048/1e6, tot["WRITE_SIZE"]*1024/1e6))
PY
