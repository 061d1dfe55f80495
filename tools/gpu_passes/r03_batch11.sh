# stream-K tail of the 8-phase GEMM: parity tests, then the whole GEMM test file, then the bench (A/B by VLLM_GEMM_SK)
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b11
timeout 900 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "stream_k" > gpurun_out/b11/pytest_sk.txt 2>&1; echo "rc $?" >> gpurun_out/b11/pytest_sk.txt
tail -15 gpurun_out/b11/pytest_sk.txt
echo skipped > gpurun_out/b11/pytest_vit.txt
tail -5 gpurun_out/b11/pytest_vit.txt
VLLM_GEMM_SK=0 timeout 600 python bench.py --workload vitl --no-cpu-baseline > gpurun_out/b11/bench_sk0.json 2> gpurun_out/b11/bench_sk0.err
timeout 600 python bench.py --workload vitl --no-cpu-baseline > gpurun_out/b11/bench_sk1.json 2> gpurun_out/b11/bench_sk1.err
python - <<'PY'
import json
for n in ("sk0", "sk1"):
    try:
        d = json.loads(open(f"gpurun_out/b11/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], {k: (v.get("us_per_launch"), v.get("frac")) for k, v in d.get("rooflines", {}).items() if "gemm" in k or "fc" in k or "qkv" in k or "proj" in k})
    except Exception as e:
        print(n, "failed", e)
PY
