set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench_c2.json 2> gpurun_out/r2a/bench_c2.err; echo "rc=$?"
python bench.py --config c4 --steps 3 --no-cpu-baseline > gpurun_out/r2a/bench_c4_n1.json 2> gpurun_out/r2a/bench_c4.err
python bench.py --config c5 --steps 3 --no-cpu-baseline > gpurun_out/r2a/bench_c5_n1.json 2> gpurun_out/r2a/bench_c5.err
python bench.py --robot kuka --batch 8192 --steps 3 --no-cpu-baseline > gpurun_out/r2a/bench_kuka_8192.json 2>/dev/null
nproc; python -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"; cat /sys/fs/cgroup/cpu.max
tail -3 gpurun_out/r2a/pytest.log
