#!/bin/bash
# round 5, pass 19: MSDA backward: phase C as a reduce-scatter over rotated points, flush through a per-round offset table
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 300 python tools/msda_bwd_variants.py > gpurun_out/r05o/bwd_variants4.txt 2>&1
cat gpurun_out/r05o/bwd_variants4.txt
