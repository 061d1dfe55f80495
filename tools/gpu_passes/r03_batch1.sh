set -x
mkdir -p gpurun_out/b1
./tools/probes/valu_issue > gpurun_out/b1/valu_issue.txt 2>&1
timeout 900 python -m pytest tests/test_vit_gpu.py -x -q -k "attention" > gpurun_out/b1/pytest_attn.txt 2>&1; echo "pytest rc $?" >> gpurun_out/b1/pytest_attn.txt
timeout 600 python tools/bench_attn.py > gpurun_out/b1/bench_attn.txt 2>&1
tail -5 gpurun_out/b1/pytest_attn.txt; cat gpurun_out/b1/bench_attn.txt; cat gpurun_out/b1/valu_issue.txt
