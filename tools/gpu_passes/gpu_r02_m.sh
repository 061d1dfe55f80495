# round 2, GPU pass M: the bench line (both workloads) + rocprofv3 kernel stats of the same command
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/r02m_bench_line.json 2> gpurun_out/r02m_bench_err.txt
tail -3 gpurun_out/r02m_bench_err.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02m_prof -o bench -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02m_bench_prof_line.json 2> gpurun_out/r02m_prof_err.txt
tail -3 gpurun_out/r02m_prof_err.txt
find gpurun_out/r02m_prof -name '*kernel_stats*' | head
f=$(find gpurun_out/r02m_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f" > gpurun_out/r02m_bench_kernel_stats.txt
find gpurun_out/r02m_prof -name '*kernel_trace*' -delete
find gpurun_out/r02m_prof -name '*.db' -delete
timeout 900 python bench.py --workload internvit6b --steps 5 --warmup 2 > gpurun_out/r02m_bench_ivit_line.json 2> gpurun_out/r02m_bench_ivit_err.txt
tail -5 gpurun_out/r02m_bench_ivit_err.txt
cat gpurun_out/r02m_bench_line.json gpurun_out/r02m_bench_ivit_line.json
