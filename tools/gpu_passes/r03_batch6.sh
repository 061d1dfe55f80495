python tools/msda_bwd_ablate.py 2>&1 | grep -v amdgpu
