# round 2, GPU pass S: LDS-tiled DCNv3 forward kernels: parity + micro-benchmark (pipelined / two-block / gather kernel) + phase clock
set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dcnv3_gpu.py -m gpu -x -q 2>&1 | tail -12
for m in 1 3 0; do echo "## dcnv3_tiled = $m"; VLLM_DCNV3_TILED=$m timeout 300 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r02s_dcnv3.txt
DCN_PROF_MODE=2 python tools/prof_dcnv3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02t_dcnv3_pipe_phases.txt
