set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b34
timeout 300 python tools/gpu_passes/dbg_lnc.py > gpurun_out/b34/dbg.txt 2>&1; tail -8 gpurun_out/b34/dbg.txt
timeout 600 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "persistent or folded_norm" > gpurun_out/b34/pytest.txt 2>&1; tail -5 gpurun_out/b34/pytest.txt
timeout 300 python tools/gemm_persist_ab.py > gpurun_out/b34/ab.txt 2>&1; tail -5 gpurun_out/b34/ab.txt
timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b34/phases.txt 2>&1; tail -8 gpurun_out/b34/phases.txt
