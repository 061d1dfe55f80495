cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
VLLM_VIT_DEBUG=1 timeout 900 python -m pytest tests/test_vit_gpu.py -x -q -s -k "folded_norms" 2>&1 | grep -E "vit:|Error|assert|passed|failed|launches" | head -20 | tee $O/pytest_fold.txt
timeout 1200 python -m pytest tests/test_vit_gpu.py tests/test_fullsize_gpu.py tests/test_ivit_depth_gpu.py -q 2>&1 | tail -6 | tee $O/pytest_vit.txt
