# round 2, GPU pass A: MSDA generation 6 (pyramid items) -- parity tests, then the micro-benchmark of every forward variant
set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_msda_gpu.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r02a_msda_tests.txt
timeout 600 python -m pytest tests -m gpu -q --deselect tests/test_msda_gpu.py 2>&1 | tail -8 | tee gpurun_out/r02a_other_tests.txt
timeout 600 python tools/bench_msda.py --iters 20 2>&1 | tail -40 | tee gpurun_out/r02a_msda_bench.txt
