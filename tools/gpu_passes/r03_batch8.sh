# full GPU suite after the q/k norm merge + rocprofv3 kernel stats of BOTH workloads
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b8
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/b8/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/b8/pytest_gpu.txt
tail -6 gpurun_out/b8/pytest_gpu.txt
rm -rf gpurun_out/b8/prof_v gpurun_out/b8/prof_i
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b8/prof_v -o bench -- python bench.py --workload vitl --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/b8/bench_prof_line_vitl.json 2> /dev/null
f=$(find gpurun_out/b8/prof_v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/b8/bench_kernel_stats_vitl.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b8/prof_i -o bench -- python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b8/bench_prof_line_internvit6b.json 2> /dev/null
f=$(find gpurun_out/b8/prof_i -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/b8/bench_kernel_stats_internvit6b.csv
find gpurun_out/b8 -name '*kernel_trace*' -delete
find gpurun_out/b8/prof_v gpurun_out/b8/prof_i -type f -size +1M -delete
head -30 gpurun_out/b8/bench_kernel_stats_internvit6b.csv | cut -c1-160
