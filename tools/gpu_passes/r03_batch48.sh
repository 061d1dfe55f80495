set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b48
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_vit_gpu.py tests/test_tokens_gpu.py -m gpu -q -x -k "persistent or folded or golden or cfg1 or two_stream or graph or tokens" > gpurun_out/b48/pytest.txt 2>&1; tail -3 gpurun_out/b48/pytest.txt
timeout 300 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms')"
