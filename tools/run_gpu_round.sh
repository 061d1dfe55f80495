set -x
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; tail -c 2500 gpurun_out/bench_r01b.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof13 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R; python tools/collect_pmc.py gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 gpurun_out/pmc_traffic2.json | head -30
python tools/rocprof_summary.py gpurun_out/prof13 2>/dev/null | head -30
