# norm kernel: weight / bias requested together with the row (side build) vs the committed kernel, same box, in the bench step
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b24
for v in base normv base normv; do
  if [ $v = normv ]; then export VLLM_HIP_LIB=$R/visionllm_amd/_build_abl/normv/libvllm_hip.so; else unset VLLM_HIP_LIB; fi
  timeout 600 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/b24/bench_$v.json 2> gpurun_out/b24/bench_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/b24/bench_$v.json").read().strip().splitlines()[-1])
print("$v:", round(d["value"], 1), "img/s", round(d["ms_per_step"], 3), "ms; norm in-step", round(d["in_step_us_per_launch"]["norm"], 2), "msda", round(d["in_step_us_per_launch"]["msda_encoder_shape"], 1))
PY
done
