cd $GRAFT_REPO_ROOT
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
python tools/gpu_passes/dbg_dcnv3_bwd.py 1 64 64 2 32 3 1 1 1 9.0 2>&1 | grep -v amdgpu
NAN=0 python tools/gpu_passes/dbg_dcnv3_bwd.py 1 64 64 2 32 3 1 1 1 9.0 2>&1 | grep -v amdgpu
python tools/gpu_passes/dbg_dcnv3_bwd.py 1 64 64 2 32 3 1 1 1 3.0 2>&1 | grep -v amdgpu
