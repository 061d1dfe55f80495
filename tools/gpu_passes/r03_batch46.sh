set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b46
SOAK_REPS=500 timeout 300 python tools/soak_gemm_persistent.py > gpurun_out/b46/soak.txt 2>&1; tail -4 gpurun_out/b46/soak.txt
