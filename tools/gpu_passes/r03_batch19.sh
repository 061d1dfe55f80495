set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b19
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -q -x > gpurun_out/b19/pytest.txt 2>&1; tail -3 gpurun_out/b19/pytest.txt
for i in 1 2; do
VLLM_HIP_LIB=$R/visionllm_amd/_build_abl/old37/libvllm_hip.so timeout 120 python tools/msda8_ab.py 2>&1 | grep "^{"
timeout 120 python tools/msda8_ab.py 2>&1 | grep "^{"
done
timeout 600 python bench.py --workload vitl --no-cpu-baseline > gpurun_out/b19/bench.json 2> gpurun_out/b19/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b19/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); r = d["rooflines"]
for k in ("msda", "msda_nonpyramid"): print(k, r[k]["kernel"][:70], round(r[k]["us_per_launch"], 1), round(r[k]["frac"], 3), round(r[k]["us_per_launch_isolated"], 1))
PY
