# round 5, pass 1: the instruction diet of msda_tiled9.hip (T9_DIET bits) -- side builds timed interleaved on one box, checked against
# the gather kernel; the MSDA test file with the library built at the new default; the race screen on the library.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python tools/msda9_variants.py 2>&1 | grep -v amdgpu | tee $O/msda9_diet.txt
timeout 900 python -m pytest tests/test_msda_gpu.py -x -q 2>&1 | tail -5 | tee $O/pytest_msda.txt
timeout 600 python tools/gpu_passes/dbg_msda9_race.py 40 2>&1 | grep -v amdgpu | tail -30 | tee $O/race_lib.txt
