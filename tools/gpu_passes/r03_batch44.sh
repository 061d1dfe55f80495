set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b44
for f in 1 0; do
VLLM_LN_FOLD=$f timeout 600 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b44/bench_$f.json 2> gpurun_out/b44/bench_$f.err
python - <<PY
import json
d = json.loads(open("gpurun_out/b44/bench_$f.json").read().strip().splitlines()[-1])
print("fold $f:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms;", {k: round(v, 1) for k, v in d["in_step_us_per_launch"].items()})
PY
done
