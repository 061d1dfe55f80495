set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 900 python -m pytest tests/test_msda_gpu.py -q -x -s -k "decoder_shape or fused_layer or skinny or module" > $O/pytest.txt 2>&1; grep -E "decoder layer|passed|failed|Error" $O/pytest.txt | tail -12
timeout 600 python -m pytest tests/test_vit_gpu.py -q -x -k "half_height" 2>&1 | tail -2
python tools/bench_msda_layer.py --case decoder 2>&1 | grep -v amdgpu | tee $O/layer_decoder.txt
VLLM_MSDA_LAYER_VALUE_BF16=0 python tools/bench_msda_layer.py --case decoder 2>&1 | grep -v amdgpu | tee -a $O/layer_decoder.txt
