set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b22
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -q -x > gpurun_out/b22/pytest.txt 2>&1; tail -3 gpurun_out/b22/pytest.txt
timeout 120 python tools/bench_msda_layer.py 2>&1 | tail -3
