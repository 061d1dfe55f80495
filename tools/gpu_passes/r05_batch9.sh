cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_region_gpu.py tests/test_tokens_gpu.py tests/test_msda_gpu.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu | tee gpurun_out/r05i/extra_rooflines.txt
import json, torch, bench
def entry(tag, kernel, bound, work, sec, n, peak, unit, scale, **kw):
    return dict(kernel=kernel[:60], us=sec*1e6, frac=work/sec/scale/peak)
r = bench.extra_rooflines("cuda:0", entry)
for k, v in r.items(): print(k, round(v["us"],1), "us", round(v["frac"],3), v["kernel"])
PY
