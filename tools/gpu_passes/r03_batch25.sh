set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "tiled_kernel_matches or generation6_pyramid or geometry_hint or layer" 2>&1 | tail -2
for i in 1 2 3; do timeout 120 python tools/msda8_ab.py 2>&1 | grep "^{"; done
