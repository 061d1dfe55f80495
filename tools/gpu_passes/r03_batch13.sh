set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b13
timeout 300 python tools/gemm_shape_sweep.py > gpurun_out/b13/gemm_shape_sweep.txt 2>&1
cat gpurun_out/b13/gemm_shape_sweep.txt
timeout 900 python -m pytest tests/test_vit_gpu.py -m gpu -q -x > gpurun_out/b13/pytest_vit.txt 2>&1; echo "rc $?" >> gpurun_out/b13/pytest_vit.txt
tail -4 gpurun_out/b13/pytest_vit.txt
