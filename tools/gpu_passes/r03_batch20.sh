set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b20
timeout 120 python tools/msda8_ab.py 2>&1 | grep -v amdgpu > gpurun_out/b20/ab.txt; cat gpurun_out/b20/ab.txt
