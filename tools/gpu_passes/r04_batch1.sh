# round 4, pass 1: generation-9 MSDA forward (correctness + A/B + phase clock), full-size parity fixtures, bench smoke with the clock sampler
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 600 python -m pytest tests/test_msda_gpu.py -q -x -k "tiled_kernel_matches or generation6_pyramid or geometry_hint or full_size_cfg4" > $O/pytest_msda9.txt 2>&1; tail -5 $O/pytest_msda9.txt
timeout 300 python tools/msda9_ab.py > $O/msda9_ab.txt 2>&1; cat $O/msda9_ab.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_capi.py -q -x > $O/pytest_fullsize.txt 2>&1; tail -15 $O/pytest_fullsize.txt
cp gpurun_out/parity_contract.jsonl $O/ 2>/dev/null
timeout 600 python bench.py --workload vitl --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_vitl.json 2> $O/bench_err.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04a/bench_vitl.json"))
print(d["value"], d["ms_per_step"], d.get("clocks"), {k:(round(v["us_per_launch"],1), round(v["frac"],3)) for k,v in d["rooflines"].items()})
PY
tail -3 $O/bench_err.txt
