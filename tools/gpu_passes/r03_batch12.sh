# stream-K tail with the measured fix-up charge: InternViT-6B and ViT-L step, A/B by VLLM_GEMM_SK
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b12
for sk in 0 1; do
  VLLM_GEMM_SK=$sk timeout 900 python bench.py --workload internvit6b --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/b12/bench_i6b_sk$sk.json 2> gpurun_out/b12/bench_i6b_sk$sk.err
  VLLM_GEMM_SK=$sk timeout 600 python bench.py --workload vitl --no-cpu-baseline > gpurun_out/b12/bench_vitl_sk$sk.json 2> gpurun_out/b12/bench_vitl_sk$sk.err
done
python - <<'PY'
import json
for n in ("i6b_sk0", "i6b_sk1", "vitl_sk0", "vitl_sk1"):
    try:
        d = json.loads(open(f"gpurun_out/b12/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], {k: v.get("us_per_launch") for k, v in d.get("rooflines", {}).items() if "gemm" in k})
    except Exception as e:
        print(n, "failed", e)
PY
