set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 900 python -m pytest tests/test_vit_gpu.py tests/test_msda_gpu.py -q -x -k "attention or flash or modules_match" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
python tools/bench_attn.py 2>&1 | tail -8 > $O/bench_attn.txt; cat $O/bench_attn.txt
