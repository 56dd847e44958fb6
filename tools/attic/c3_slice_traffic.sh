# c3 (table scene, workgroup kernel): solves/s and HBM traffic of the solve kernel by time-slice length
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SL in 96 160 256; do
  export GIK_SLICE=$SL
  python $R/bench.py --config c3 --headline-only --no-cpu-baseline --serving-streams 0 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slice $SL', round(d['value'],1), 'solves/s', round(d['roofline']['kernel_ms'],1), 'ms kernel')"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_$C
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pm_$C -o r1 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --serving-streams 0 --headline-only --config c3 > /dev/null 2> /tmp/pm.err
    python - <<PY
import csv,glob
f=glob.glob("/tmp/pm_$C/**/r1_counter_collection.csv", recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="$C" and "rtr_block" in r["Kernel_Name"]]
print("  slice $SL $C KiB per dispatch", [round(x) for x in v], "-> MB", round(sum(v)/len(v)*1024*(2 if "$C"=="FETCH_SIZE" else 1)/1e6,1))
PY
  done
done
