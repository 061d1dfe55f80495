#!/bin/bash
# round 5, pass 23: backward tests + DCNv3 backward + phase clock on the restructured kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 1200 python -m pytest tests/test_msda_gpu.py tests/test_dcnv3_gpu.py tests/test_race_screen_gpu.py -x -q -k "backward or bwd or race or grad or autograd" 2>&1 | tail -8 > gpurun_out/r05o/bwd_tests.txt
cat gpurun_out/r05o/bwd_tests.txt
for lib in _build; do
  echo "== $lib"
  DCN_BWD=1 VLLM_HIP_LIB=$GRAFT_REPO_ROOT/visionllm_amd/$lib/libvllm_hip.so timeout 300 python tools/bench_dcnv3.py 2>&1 | grep backward
done > gpurun_out/r05o/dcn_bwd_tiles.txt 2>&1
cat gpurun_out/r05o/dcn_bwd_tiles.txt
timeout 120 python tools/msda_bwd_phases.py libprof_mfma_w8b3.so 2>&1 | tee gpurun_out/r05o/bwd_phases_final.txt
