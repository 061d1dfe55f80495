#!/bin/bash
# round 5, pass 33: soak -- the race screen three times over + the backward tests twice (the rebuilt backward, the native splice)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
for i in 1 2 3; do timeout 600 python -m pytest tests/test_race_screen_gpu.py -x -q 2>&1 | tail -1; done | tee gpurun_out/r05o/soak.txt
for i in 1 2; do timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_dcnv3_gpu.py tests/test_tokens_gpu.py -x -q -k "backward or bwd or grad or splice or token" 2>&1 | tail -1; done | tee -a gpurun_out/r05o/soak.txt
