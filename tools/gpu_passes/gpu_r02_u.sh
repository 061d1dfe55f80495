# round 2, GPU pass U: does the SLP vectoriser (v_pk_* f32 pairs) cost time in the MFMA kernels?  default build vs -fno-slp-vectorize
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  python tools/ab_libs.py 2>&1 | grep "\"lib\""
  VLLM_HIP_LIB=$GRAFT_REPO_ROOT/visionllm_amd/_build_noslp/libvllm_hip.so python tools/ab_libs.py 2>&1 | grep "\"lib\""
done | tee gpurun_out/r02u_slp_ab.txt
