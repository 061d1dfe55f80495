# generation 8 vs generation 7 inside the bench step, same box
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b18
for m in 15 1 15 1; do
  VLLM_MSDA_TILED=$m timeout 600 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b18/bench_$m.json 2> gpurun_out/b18/bench_$m.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/b18/bench_$m.json").read().strip().splitlines()[-1])
r = d["rooflines"]["msda"]
print("mode $m:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms; msda in-step", round(r["us_per_launch"], 1), "isolated", round(r["us_per_launch_isolated"], 1), "msda_12_calls", round(d["phases_ms"]["msda_12_calls"], 3))
PY
done
