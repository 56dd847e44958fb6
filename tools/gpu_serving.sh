mkdir -p gpurun_out/serv
for S in 4 8; do python bench.py --streams $S --steps 16 --serving-streams 0 --no-cpu-baseline > gpurun_out/serv/streams_$S.json 2>/dev/null; done
python bench.py --serving-streams 8 --no-cpu-baseline > gpurun_out/serv/serving8.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --serving-streams 8 --no-cpu-baseline > gpurun_out/serv/serving8_q8.json 2>/dev/null
GPU_MAX_HW_QUEUES=8 python bench.py --streams 4 --steps 16 --serving-streams 0 --no-cpu-baseline > gpurun_out/serv/streams_4_q8.json 2>/dev/null
for f in gpurun_out/serv/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), d["ms_per_step"], d.get("serving",{}).get("value"))
PY
done
