# round 5, pass 2: diet steps 16 / 32 (contiguous DMA runs, layout fast path), rebalancing options on top; the new race-screen test file
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python tools/msda9_variants.py 2>&1 | grep -v amdgpu | tee $O/msda9_diet.txt
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_race_screen_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest_msda.txt
