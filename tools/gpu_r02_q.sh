# round 2, GPU pass Q: generation 7 with the conflict-free pair gather: parity, micro-benchmark, phase clock
set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "generation6 or tiled_kernel" 2>&1 | tail -5
timeout 300 python tools/bench_msda.py --iters 20 --modes encoder_like 2>&1 | grep "f32_auto\|gen7" | grep "37485" | tee gpurun_out/r02q_msda_pair.txt
T6_PROF_MODE=19 timeout 300 python tools/prof_msda6.py 2>&1 | tail -25 | tee gpurun_out/r02q_msda7_pair_phases.txt
