#!/bin/bash
# round 5, pass 24: ablations + phase clock of the final backward kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 300 python tools/msda_bwd_variants.py > gpurun_out/r05o/bwd_final_ablation.txt 2>&1
timeout 120 python tools/msda_bwd_phases.py libprof_mfma_w8b3.so >> gpurun_out/r05o/bwd_final_ablation.txt 2>&1
grep " ms\| %\|level passes" gpurun_out/r05o/bwd_final_ablation.txt
