#!/bin/bash
# round 5, pass 29: DPP wave sums in the norm kernels: tests + the InternViT-6B q/k-norm in the step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 900 python -m pytest tests/test_vit_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | tail -4
timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']
print('internvit6b', d['value'], d['ms_per_step'], 'qk_norm', r['qk_norm']['us_per_launch'], r['qk_norm']['frac'], 'norm', r['norm']['us_per_launch'], r['norm']['frac'])" | tee gpurun_out/r05o/norm_dpp.txt
