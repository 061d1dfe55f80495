set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 900 python -m pytest tests/test_dcnv3_gpu.py -q -x > $O/pytest_dcnv3.txt 2>&1; tail -25 $O/pytest_dcnv3.txt
