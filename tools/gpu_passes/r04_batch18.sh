cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O
python tools/msda9_variants.py 2>&1 | grep -v amdgpu | tee $O/msda9_early.txt
for n in e1 e2; do MSDA9_LIB=$PWD/visionllm_amd/_build_abl/libmsda9_$n.so timeout 600 python tools/gpu_passes/dbg_msda9_race.py 2>&1 | grep -v amdgpu | tail -8 | tee -a $O/race_$n.txt; done
