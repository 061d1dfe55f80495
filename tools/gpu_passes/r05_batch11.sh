# round 5: GEMM main-loop experiment -- the phase-4 counted wait behind the MFMA section (P_LATE_WAIT); two library builds, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
for i in 1 2 3; do
  python tools/ab_libs.py 2>/dev/null | tail -1 | tee -a $O/gemm_late_wait.txt
  VLLM_HIP_LIB=$PWD/visionllm_amd/_build_late/libvllm_hip.so python tools/ab_libs.py 2>/dev/null | tail -1 | tee -a $O/gemm_late_wait.txt
done
VLLM_HIP_LIB=$PWD/visionllm_amd/_build_late/libvllm_hip.so timeout 600 python -m pytest tests/test_vit_gpu.py -q -x -k "gemm" 2>&1 | tail -3 | tee $O/pytest_gemm_late.txt
