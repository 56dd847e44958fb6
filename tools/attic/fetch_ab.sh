cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in new old; do
  if [ $L = old ]; then export GIK_LIB_PATH=$R/graphik_amd/lib/exp/libgraphik_amd_old.so; fi
  rm -rf /tmp/pm_$L
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pm_$L -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serving-streams 0 --headline-only --config c5 > /dev/null 2> /tmp/pm_$L.err
  python - <<PY
import csv,glob
f=glob.glob("/tmp/pm_$L/**/r1_counter_collection.csv", recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "rtr_wave" in r["Kernel_Name"]]
print("$L", [round(x) for x in v])
PY
done
