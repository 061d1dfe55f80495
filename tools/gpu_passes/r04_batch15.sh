set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 600 python -m pytest tests/test_vit_gpu.py -q -x -k "half_height or persistent" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
timeout 900 python tools/gemm_half_tail_ab.py 2>&1 | grep -v amdgpu > $O/half_tail_ab.txt; cat $O/half_tail_ab.txt
