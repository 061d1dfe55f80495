cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python tools/msda9_variants.py 2>&1 | grep -v amdgpu | tee $O/msda9_early.txt
