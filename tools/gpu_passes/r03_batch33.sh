set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b33
timeout 300 python tools/gpu_passes/dbg_lnc.py > gpurun_out/b33/dbg.txt 2>&1; tail -30 gpurun_out/b33/dbg.txt
timeout 300 python tools/gemm_persist_ab.py --phases > gpurun_out/b33/phases.txt 2>&1; tail -10 gpurun_out/b33/phases.txt
