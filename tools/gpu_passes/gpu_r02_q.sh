# round 2, GPU pass Q: generation 7 gather variants: parity, micro-benchmark, phase clock
set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "generation6 or tiled_kernel or full_size" 2>&1 | tail -5
timeout 300 python tools/bench_msda.py --iters 20 --modes encoder_like,stress 2>&1 | grep "f32_auto\|gen" | grep "37485" | tee gpurun_out/r02q_msda.txt
T6_PROF_MODE=16 timeout 300 python tools/prof_msda6.py 2>&1 | tail -25 | tee gpurun_out/r02q_msda7_phases.txt
