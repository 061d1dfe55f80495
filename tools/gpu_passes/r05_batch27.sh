#!/bin/bash
# round 5, pass 27: DCNv3 backward with the windowed kernel at 3 (default) and 2 blocks per CU (the DCN instantiation spills 19 registers at 168)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
for lib in _build _build_b2 _build _build_b2; do
  echo "== $lib"
  DCN_BWD=1 VLLM_HIP_LIB=$GRAFT_REPO_ROOT/visionllm_amd/$lib/libvllm_hip.so timeout 300 python tools/bench_dcnv3.py 2>&1 | grep backward | cut -c1-90
done 2>&1 | tee gpurun_out/r05o/dcn_bwd_b2.txt
