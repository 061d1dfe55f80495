# round 2, GPU pass N: fused MSDA layer (tests + A/B timing), rocprofv3 kernel stats of bench.py (csv), HBM traffic PMC passes
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "layer or prep or f32" 2>&1 | tail -8 | tee gpurun_out/r02n_layer_tests.txt
timeout 300 python tools/bench_msda_layer.py 2>&1 | tail -4 | tee gpurun_out/r02n_msda_layer.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02n_prof -o bench -- python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02n_bench_prof_line.json 2> gpurun_out/r02n_prof_err.txt
f=$(find gpurun_out/r02n_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f" > gpurun_out/r02n_bench_kernel_stats.csv
find gpurun_out/r02n_prof -name '*kernel_trace*' -delete
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/r02n_fetch gpurun_out/r02n_write
(cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02n_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02n_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/collect_pmc.py gpurun_out/r02n_fetch gpurun_out/r02n_write gpurun_out/r02n_pmc_traffic.json vitl | head -40
find gpurun_out/r02n_fetch gpurun_out/r02n_write -name '*.csv' -size +2M -delete
