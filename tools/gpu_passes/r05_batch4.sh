cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python tools/msda9_variants.py 2>&1 | grep -v amdgpu | tee $O/msda9_cur.txt
timeout 300 python tools/prof_msda9.py 2>&1 | grep -v amdgpu | tee $O/msda9_phases.txt
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_race_screen_gpu.py tests/test_dcnv3_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
