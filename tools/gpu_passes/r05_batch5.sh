cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python tools/gpu_passes/r05_attn_epi.py 2>&1 | grep -v amdgpu | tee $O/attn_epi.txt
timeout 600 python -m pytest tests/test_dcnv3_gpu.py -x -q 2>&1 | tail -4 | tee $O/pytest_dcnv3.txt
