set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 900 python -m pytest tests/test_msda_gpu.py -q -x -k "skinny or fused_layer or msda_layer or module" > $O/pytest_skinny.txt 2>&1; tail -15 $O/pytest_skinny.txt
timeout 600 python tools/bench_msda_layer.py > $O/msda_layer.txt 2>&1; tail -30 $O/msda_layer.txt
