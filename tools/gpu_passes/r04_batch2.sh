# round 4, pass 2: generation 9 after the interleaved multiply-add chains (also in generation 8), priority / ablation side builds
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 600 python -m pytest tests/test_msda_gpu.py -q -x -k "tiled_kernel_matches or generation6_pyramid or geometry_hint or full_size_cfg4" > $O/pytest_msda9.txt 2>&1; tail -3 $O/pytest_msda9.txt
timeout 300 python tools/msda9_ab.py > $O/msda9_ab.txt 2>&1; cat $O/msda9_ab.txt | grep -v amdgpu.ids
timeout 600 python tools/msda9_variants.py > $O/msda9_variants.txt 2>&1; cat $O/msda9_variants.txt | grep -v amdgpu.ids
