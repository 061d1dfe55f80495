set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b43
timeout 300 python tools/gpu_passes/dbg_res.py > gpurun_out/b43/dbg.txt 2>&1; grep -v "^  " gpurun_out/b43/dbg.txt | tail -9
timeout 900 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "persistent" > gpurun_out/b43/pytest.txt 2>&1; tail -6 gpurun_out/b43/pytest.txt
timeout 300 python tools/gemm_persist_ab.py > gpurun_out/b43/ab.txt 2>&1; grep "proj\|fc2" gpurun_out/b43/ab.txt
