set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b19
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -q -x > gpurun_out/b19/pytest.txt 2>&1; tail -3 gpurun_out/b19/pytest.txt
for i in 1 2; do
VLLM_HIP_LIB=$R/visionllm_amd/_build_abl/old37/libvllm_hip.so timeout 120 python tools/msda8_ab.py 2>&1 | grep "^{"
timeout 120 python tools/msda8_ab.py 2>&1 | grep -v amdgpu > gpurun_out/b19/ab.txt; grep "^{" gpurun_out/b19/ab.txt
done
cat gpurun_out/b19/ab.txt
