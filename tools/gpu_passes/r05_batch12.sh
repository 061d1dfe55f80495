cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
timeout 600 python tools/msda9_variants.py 2>&1 | grep -v amdgpu | tee $O/msda9_cold.txt
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_race_screen_gpu.py -q -x 2>&1 | tail -4 | tee $O/pytest_msda.txt
