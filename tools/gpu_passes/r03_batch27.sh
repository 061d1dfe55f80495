set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b27
timeout 300 python tools/ln_fold_ab.py > gpurun_out/b27/ab.txt 2>&1; tail -8 gpurun_out/b27/ab.txt
timeout 600 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "folded" > gpurun_out/b27/pytest.txt 2>&1; tail -8 gpurun_out/b27/pytest.txt
