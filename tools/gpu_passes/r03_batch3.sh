set -x
mkdir -p gpurun_out/b3
timeout 900 python -m pytest tests/test_vit_gpu.py -x -q -k "attention" > gpurun_out/b3/pytest_attn.txt 2>&1; echo "pytest rc $?" >> gpurun_out/b3/pytest_attn.txt
tail -5 gpurun_out/b3/pytest_attn.txt
timeout 600 python tools/bench_attn.py 64,1088,192 > gpurun_out/b3/bench_attn.txt 2>&1
cat gpurun_out/b3/bench_attn.txt
