# generation 7 on nested (ceil-divided) level maps: parity + A/B of the bench shapes
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b16
timeout 900 python -m pytest tests/test_msda_gpu.py -m gpu -q -x > gpurun_out/b16/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/b16/pytest.txt
tail -5 gpurun_out/b16/pytest.txt
timeout 300 python tools/msda8_ab.py 2>&1 | grep -v amdgpu | head -4 > gpurun_out/b16/ab.txt; cat gpurun_out/b16/ab.txt
timeout 600 python bench.py --workload vitl --no-cpu-baseline > gpurun_out/b16/bench.json 2> gpurun_out/b16/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b16/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); r = d["rooflines"]
for k in ("msda", "msda_nonpyramid"): print(k, r[k]["kernel"][:70], round(r[k]["us_per_launch"], 1), round(r[k]["frac"], 3))
PY
