set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b31
timeout 600 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "persistent or folded_norm" > gpurun_out/b31/pytest.txt 2>&1; tail -12 gpurun_out/b31/pytest.txt
timeout 300 python tools/gemm_persist_ab.py > gpurun_out/b31/ab.txt 2>&1; tail -6 gpurun_out/b31/ab.txt
timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b31/phases.txt 2>&1; tail -10 gpurun_out/b31/phases.txt
