cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python bench.py --workload vitl > $O/bench_vitl.json 2> $O/bench_err.txt; tail -c 400 $O/bench_err.txt
python - <<PY
import json
d=json.load(open("$O/bench_vitl.json"))
print(d["value"], "img/s", d["ms_per_step"], "ms", d.get("clocks"))
print({k: round(v,1) for k,v in d.get("in_step_us_per_launch",{}).items()})
for k,v in d["rooflines"].items():
    print(k, {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("frac","us_per_launch","us_per_launch_isolated","error","launches_per_step")})
print(json.dumps(d["cpu_baseline"])[:900])
PY
