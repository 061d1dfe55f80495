set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b15
timeout 600 python -m pytest tests/test_vit_gpu.py -m gpu -q -x -k "norm or hook or layer or encoder or vit" > gpurun_out/b15/pytest.txt 2>&1; echo "rc $?" >> gpurun_out/b15/pytest.txt
tail -4 gpurun_out/b15/pytest.txt
timeout 600 python bench.py --workload vitl --no-cpu-baseline > gpurun_out/b15/bench.json 2> gpurun_out/b15/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b15/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["in_step_us_per_launch"], d["rooflines"]["norm"]["us_per_launch_isolated"])
PY
