# One GPU-box pass that regenerates everything profiles/ quotes: full -m gpu suite, smoke, default bench line, rocprofv3 kernel
# stats of the bench, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only), micro-benchmarks.
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; tail -c 600 gpurun_out/bench_r01b.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof13 $R/gpurun_out/pmc_fetch2 $R/gpurun_out/pmc_write2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof13 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; python tools/collect_pmc.py gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 gpurun_out/pmc_traffic2.json | head -30
python tools/rocprof_summary.py gpurun_out/prof13 2>/dev/null | head -34 | tee gpurun_out/kernel_stats.txt
python tools/bench_attn.py 2>&1 | grep "variant 32" | tee gpurun_out/bench_attn.txt
python tools/gemm_ab.py 2>&1 | tee gpurun_out/gemm_ab.txt
python tools/bench_msda.py 2>&1 | tail -12 | tee gpurun_out/msda_bench.txt
