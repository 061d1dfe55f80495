#!/bin/bash
# round 5, pass 18: coalesced per-point gradient stores; plain-store flush ablation
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 300 python tools/msda_bwd_variants.py > gpurun_out/r05o/bwd_variants3.txt 2>&1
cat gpurun_out/r05o/bwd_variants3.txt
