set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
DCN_BWD=1 timeout 600 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu > $O/dcnv3_bwd.txt; cat $O/dcnv3_bwd.txt
DCN_BWD=1 DCN_C16=1 timeout 600 python tools/bench_dcnv3.py 2>&1 | grep -v amdgpu > $O/dcnv3_bwd_c16.txt; cat $O/dcnv3_bwd_c16.txt
