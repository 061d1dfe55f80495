set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 900 python -m pytest tests/test_dcnv3_gpu.py -q -x -k "backward" > $O/pytest_dcnv3.txt 2>&1; tail -25 $O/pytest_dcnv3.txt
timeout 600 python -m pytest tests/test_msda_gpu.py -q -x -k "backward or bwd or grad" > $O/pytest_msda_bwd.txt 2>&1; tail -5 $O/pytest_msda_bwd.txt
DCN_BWD=1 timeout 600 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu > $O/dcnv3_bwd.txt; cat $O/dcnv3_bwd.txt
