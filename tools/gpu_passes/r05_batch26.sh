#!/bin/bash
# round 5, pass 26: DCNv3 locations on / next to integers in every kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 600 python -m pytest tests/test_dcnv3_gpu.py -x -q -k "integers" 2>&1 | tail -30
