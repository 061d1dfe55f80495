set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b32
timeout 600 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "persistent or folded_norm" > gpurun_out/b32/pytest.txt 2>&1; tail -12 gpurun_out/b32/pytest.txt
timeout 300 python tools/gemm_persist_ab.py > gpurun_out/b32/ab.txt 2>&1; tail -6 gpurun_out/b32/ab.txt
timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b32/phases.txt 2>&1; tail -10 gpurun_out/b32/phases.txt
VLLM_GEMM_PROF=3 timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b32/phases_nostore.txt 2>&1; tail -10 gpurun_out/b32/phases_nostore.txt
