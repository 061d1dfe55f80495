set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b23
timeout 300 python tools/graph_capture_check.py 2>&1 | grep -v amdgpu | tail -8 > gpurun_out/b23/graph.txt; cat gpurun_out/b23/graph.txt
