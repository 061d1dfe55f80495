#!/bin/bash
# round 5, pass 31: attention: ragged last query tiles scheduled last (A/B, bit-identical) + the attention tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 600 python tools/gpu_passes/r05_attn_tail.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05o/attn_tail.txt
timeout 900 python -m pytest tests/test_vit_gpu.py -x -q -k "attention or attn or flash" 2>&1 | tail -3
