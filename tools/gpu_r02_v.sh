cd $GRAFT_REPO_ROOT
for r in 1 2; do for s in 0 1 2 3; do echo "stagger $s: $(VLLM_GEMM_STAGGER=$s python tools/ab_libs.py 2>&1 | grep '"lib"')"; done; done | tee gpurun_out/r02v_gemm_stagger.txt
