set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b40
timeout 1500 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "gemm" > gpurun_out/b40/pytest.txt 2>&1; tail -5 gpurun_out/b40/pytest.txt
timeout 300 python tools/gemm_persist_ab.py > gpurun_out/b40/ab.txt 2>&1; tail -5 gpurun_out/b40/ab.txt
VLLM_GEMM_PERSIST=1 timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b40/bench_ivit_1.json 2> gpurun_out/b40/bench_ivit_1.err
VLLM_GEMM_PERSIST=0 timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b40/bench_ivit_0.json 2> gpurun_out/b40/bench_ivit_0.err
python - <<PY
import json
for f in (0, 1):
    d = json.loads(open(f"gpurun_out/b40/bench_ivit_{f}.json").read().strip().splitlines()[-1])
    print("ivit persist", f, round(d["value"], 2), "img/s", round(d["ms_per_step"], 2), "ms;", {k: round(v, 1) for k, v in d["in_step_us_per_launch"].items()})
PY
