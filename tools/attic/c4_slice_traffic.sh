R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for tag in s2048 noslice; do
  if [ $tag = s2048 ]; then export GIK_SLICE=2048; unset GIK_DBG; else unset GIK_SLICE; export GIK_DBG=1024; fi
  P=$R/gpurun_out/traffic_$tag; mkdir -p $P
  for set in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pmc_$set -o r1 -- python $R/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --serving-streams 0 --headline-only > $P/bench_$set.json 2> $P/err_$set.txt
  done
done
