# full -m gpu suite + smoke + the default bench (both workloads in one line) -- correctness gate after the round-3 host changes
set -x
mkdir -p gpurun_out/b5
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/b5/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/b5/pytest_gpu.txt
tail -8 gpurun_out/b5/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
( time timeout 900 python bench.py > gpurun_out/b5/bench_line.json 2> gpurun_out/b5/bench_err.txt ) 2>&1 | tail -3
tail -c 1500 gpurun_out/b5/bench_err.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/b5/bench_line.json"))
print("value", d["value"], "ms/step", d["ms_per_step"])
for k, v in d["rooflines"].items():
    if "us_per_launch" in v:
        print(f"  {k:16s} in-step {v['us_per_launch']:8.1f} us  isolated {v.get('us_per_launch_isolated', float('nan')):8.1f} us  frac {v['frac'] if v['frac'] is None else round(v['frac'], 3)}  x{v['launches_per_step']}")
print("in_step", d.get("in_step_us_per_launch"))
i = d.get("internvit6b")
if i:
    print("internvit6b value", i["value"], "ms/step", i["ms_per_step"])
    for k, v in i["rooflines"].items():
        if "us_per_launch" in v:
            print(f"  {k:16s} in-step {v['us_per_launch']:8.1f} us  isolated {v.get('us_per_launch_isolated', float('nan')):8.1f} us  frac {v['frac'] if v['frac'] is None else round(v['frac'], 3)}")
PY
