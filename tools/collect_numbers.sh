# tools/collect_numbers.sh -- every BASELINE config at its per-GPU size + the large-batch / serving
# points quoted in DESIGN.md section 6 (run on the GPU box: gpurun -- 'bash tools/collect_numbers.sh')
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err
python bench.py --config c3 --steps 2 --warmup 1 > gpurun_out/final/bench_c3.json 2>/dev/null
python bench.py --config c3 --intended > gpurun_out/final/bench_c3_intended.json 2>/dev/null
python bench.py --config c4 --steps 3 --no-cpu-baseline > gpurun_out/final/bench_c4_n1.json 2>/dev/null
python bench.py --config c5 --steps 3 --no-cpu-baseline > gpurun_out/final/bench_c5_n1.json 2>/dev/null
for cfg in "lwa4d 16384" "kuka 4096" "kuka 8192" "ur10 4096" "planar10 4096" "planar10 8192" "planar10_halfpi 4096"; do set -- $cfg; python bench.py --robot $1 --batch $2 --steps 3 --no-cpu-baseline > gpurun_out/final/bench_$1_$2.json 2>/dev/null; done
python -u tools/attic/dev_single_goal.py > gpurun_out/final/single_goal.txt 2>&1
for f in gpurun_out/final/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d["value"]), "ms", round(d["ms_per_step"],2), "kernel", round(d["roofline"]["kernel_ms"],2), "frac", round(d["roofline"]["frac"],4), "succ", round(d["success_rate"],4), "maxit", round(d["frac_maxiter"],4), "serving", round(d.get("serving",{}).get("value",0)), "cpu", round(d.get("cpu_baseline",{}).get("value",0)))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
tail -3 gpurun_out/final/single_goal.txt
