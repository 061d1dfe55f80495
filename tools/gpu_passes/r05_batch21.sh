#!/bin/bash
# round 5, pass 21: what the flush costs: half the atomics, all inside 1 MiB, plain stores
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 300 python tools/msda_bwd_variants.py > gpurun_out/r05o/bwd_flush_ablation.txt 2>&1
cat gpurun_out/r05o/bwd_flush_ablation.txt
