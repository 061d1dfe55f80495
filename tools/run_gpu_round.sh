# One GPU-box pass that regenerates what profiles/r02_* quotes (round 2): full -m gpu suite, smoke, the bench line of both
# workloads, rocprofv3 kernel stats of the bench (csv), FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only),
# the micro-benchmarks and phase clocks.  The individual passes of the round, as they were run, are in tools/gpu_passes/.
#   gpurun --timeout 2400 -- 'bash tools/run_gpu_round.sh'      then copy gpurun_out/r02z_* into profiles/ under their r02_ names
set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r02z_bench_line.json 2> gpurun_out/r02z_bench_err.txt; tail -c 400 gpurun_out/r02z_bench_line.json
timeout 900 python bench.py --workload internvit6b --steps 5 --warmup 2 > gpurun_out/r02z_bench_line_internvit6b.json 2>> gpurun_out/r02z_bench_err.txt
rm -rf gpurun_out/r02z_prof gpurun_out/r02z_fetch gpurun_out/r02z_write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02z_prof -o bench -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02z_bench_prof_line.json 2> /dev/null
f=$(find gpurun_out/r02z_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/r02z_bench_kernel_stats.csv
find gpurun_out/r02z_prof -name '*kernel_trace*' -delete
(cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02z_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02z_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/collect_pmc.py gpurun_out/r02z_fetch gpurun_out/r02z_write gpurun_out/r02z_pmc_traffic.json vitl | head -30
find gpurun_out/r02z_fetch gpurun_out/r02z_write -name '*.csv' -size +2M -delete
python tools/bench_msda.py --iters 20 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_msda_microbench.txt
T6_PROF_MODE=16 python tools/prof_msda6.py 2>&1 | tail -22 | tee gpurun_out/r02z_msda7_phases.txt
python tools/bench_msda_layer.py 2>&1 | tail -2 | tee gpurun_out/r02z_msda_layer.txt
python tools/bench_attn.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_attn.txt
python tools/attn_zero_data.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_attn_zero_data.txt
python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_dcnv3_tiled.txt
python tools/prof_dcnv3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_dcnv3_phases.txt
python tools/prof_gemm256.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_gemm256_phases.txt
python tools/trace_gemm256.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02z_gemm256_block_trace.txt
