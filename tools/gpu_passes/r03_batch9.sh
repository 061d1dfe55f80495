# generation 8 of the MSDA forward (two teams half a period apart): parity tests + A/B against generation 7 + phase clock
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b9
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "tiled_kernel_matches or generation6_pyramid" > gpurun_out/b9/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/b9/pytest.txt
tail -15 gpurun_out/b9/pytest.txt
timeout 300 python tools/msda8_ab.py > gpurun_out/b9/ab.txt 2>&1; echo "rc $?" >> gpurun_out/b9/ab.txt
cat gpurun_out/b9/ab.txt
