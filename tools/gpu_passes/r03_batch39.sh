set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b39
timeout 1500 python -m pytest tests/test_vit_gpu.py tests/test_capi.py -m gpu -q -x > gpurun_out/b39/pytest.txt 2>&1; tail -5 gpurun_out/b39/pytest.txt
for f in 0 1; do
VLLM_GEMM_PERSIST=$f timeout 600 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b39/bench_$f.json 2> gpurun_out/b39/bench_$f.err
python - <<PY
import json
d = json.loads(open("gpurun_out/b39/bench_$f.json").read().strip().splitlines()[-1])
print("persist $f:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms;", {k: round(v, 1) for k, v in d["in_step_us_per_launch"].items()})
PY
done
VLLM_GEMM_PERSIST=1 timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b39/bench_ivit_1.json 2> gpurun_out/b39/bench_ivit_1.err
VLLM_GEMM_PERSIST=0 timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/b39/bench_ivit_0.json 2> gpurun_out/b39/bench_ivit_0.err
python - <<PY
import json
for f in (0, 1):
    d = json.loads(open(f"gpurun_out/b39/bench_ivit_{f}.json").read().strip().splitlines()[-1])
    print("ivit persist", f, round(d["value"], 2), "img/s", round(d["ms_per_step"], 2), "ms;", {k: round(v, 1) for k, v in d["in_step_us_per_launch"].items()})
PY
