mkdir -p gpurun_out/prio
run() { # tag robot batch env...
  tag=$1; robot=$2; batch=$3; shift 3
  env "$@" python bench.py --robot $robot --batch $batch --steps 3 --serving-streams 0 --no-cpu-baseline > gpurun_out/prio/${robot}_${batch}_$tag.json 2>/dev/null
  python - gpurun_out/prio/${robot}_${batch}_$tag.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d["value"]), "ms", round(d["ms_per_step"],1), "kernel", round(d["roofline"]["kernel_ms"],1))
PY
}
for cfg in "lwa4d 4096" "lwa4d 8192" "lwa4d 16384" "kuka 8192" "kuka 65536"; do
  set -- $cfg
  run w4 $1 $2 GIK_WAVES_PER_CU=4
  run w8 $1 $2 GIK_WAVES_PER_CU=8
  run w4p $1 $2 GIK_WAVES_PER_CU=4 GIK_DBG=64
  run w8p $1 $2 GIK_WAVES_PER_CU=8 GIK_DBG=64
done
