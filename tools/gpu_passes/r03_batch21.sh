set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b21
timeout 300 python tools/msda8_ablate.py 2>&1 | grep -v amdgpu > gpurun_out/b21/abl.txt; cat gpurun_out/b21/abl.txt
