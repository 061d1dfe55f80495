set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b42
timeout 300 python tools/gpu_passes/dbg_res.py > gpurun_out/b42/dbg.txt 2>&1; tail -30 gpurun_out/b42/dbg.txt
