set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b37
timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b37/phases.txt 2>&1; grep persistent gpurun_out/b37/phases.txt
VLLM_GEMM_PROF=3 timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b37/phases_nostore.txt 2>&1; grep persistent gpurun_out/b37/phases_nostore.txt
