#!/bin/bash
# round 5, pass 15: whole-tensor digests of the full-size fixtures (first run: calibrates the bound) + MSDA backward tile / occupancy variants
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -s 2>&1 | tail -40 > gpurun_out/r05o/fullsize.txt
timeout 300 python tools/msda_bwd_variants.py > gpurun_out/r05o/bwd_variants.txt 2>&1
for n in libprof_mfma_w16b2.so libprof_mfma_w8b3.so; do echo "== $n"; timeout 120 python tools/msda_bwd_phases.py $n; done > gpurun_out/r05o/bwd_phases.txt 2>&1
cat gpurun_out/r05o/bwd_variants.txt gpurun_out/r05o/bwd_phases.txt; tail -5 gpurun_out/r05o/fullsize.txt
