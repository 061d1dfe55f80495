set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b17
timeout 300 python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "tiled_kernel_matches or generation6_pyramid" > gpurun_out/b17/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/b17/pytest.txt
tail -6 gpurun_out/b17/pytest.txt
timeout 120 python tools/msda8_ab.py 2>&1 | grep -v amdgpu > gpurun_out/b17/ab.txt; cat gpurun_out/b17/ab.txt
