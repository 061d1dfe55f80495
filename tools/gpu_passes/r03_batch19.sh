set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b19
for i in 1 2; do
VLLM_HIP_LIB=$R/visionllm_amd/_build_abl/old37/libvllm_hip.so timeout 120 python tools/msda8_ab.py 2>&1 | grep "^{"
timeout 120 python tools/msda8_ab.py 2>&1 | grep "^{"
done
