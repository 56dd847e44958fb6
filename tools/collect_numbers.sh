mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
for cfg in "lwa4d 16384" "lwa4d 65536" "kuka 4096" "kuka 8192" "ur10 4096" "planar10 4096" "planar10 8192" "planar10 16384" "planar10_halfpi 4096"; do set -- $cfg; python bench.py --robot $1 --batch $2 --steps 3 --no-cpu-baseline > gpurun_out/final/bench_$1_$2.json 2>/dev/null; done
python -u tools/dev_table_time.py > gpurun_out/final/table.txt 2>&1
python -u tools/dev_latency.py > gpurun_out/final/latency.txt 2>&1
python -u tools/dev_segtime.py >> gpurun_out/final/latency.txt 2>&1
python bench.py --robot ur10_table --batch 4096 --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/bench_ur10_table_4096.json 2>/dev/null
