set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b28
timeout 300 python tools/ln_fold_ab.py --phases > gpurun_out/b28/phases.txt 2>&1; tail -10 gpurun_out/b28/phases.txt
