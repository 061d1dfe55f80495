# round 2, GPU pass S: LDS-tiled DCNv3 forward: parity + micro-benchmark (tiled vs gather kernel)
set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_dcnv3_gpu.py -m gpu -x -q 2>&1 | tail -12
timeout 300 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02s_dcnv3_tiled.txt
VLLM_DCNV3_TILED=0 timeout 300 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02s_dcnv3_gather.txt
DCN_OFFSET_SIGMA=0.3 timeout 300 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02s_dcnv3_tiled_sigma03.txt
