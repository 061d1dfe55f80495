set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_msda_gpu.py -q -x > $O/pytest_msda.txt 2>&1; tail -5 $O/pytest_msda.txt
python tools/msda8_ab.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/msda_ab.txt
FRESH=1 timeout 900 python tools/gpu_passes/dbg_msda9_race.py 2>&1 | grep -v amdgpu | tail -24 | tee $O/race_lib.txt
