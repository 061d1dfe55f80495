set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b41
timeout 900 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "persistent" > gpurun_out/b41/pytest.txt 2>&1; tail -12 gpurun_out/b41/pytest.txt
timeout 300 python tools/gemm_persist_ab.py > gpurun_out/b41/ab.txt 2>&1; tail -8 gpurun_out/b41/ab.txt
timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b41/phases.txt 2>&1; tail -14 gpurun_out/b41/phases.txt
