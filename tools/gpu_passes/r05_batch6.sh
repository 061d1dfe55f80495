# round 5, pass 6: InternViT-6B norm fold (wide RMS statistics) -- the folded-vs-launched test at hidden 3200, the full-size contract
# tests, and the configs[2] step with the fold on / off on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests/test_vit_gpu.py -x -q -k "folded_norms or persistent or half_height" 2>&1 | tail -6 | tee $O/pytest_fold.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ivit_depth_gpu.py -x -q 2>&1 | tail -6 | tee $O/pytest_fullsize.txt
for f in 1 0; do
  VLLM_LN_FOLD_WIDE=$f timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_ivit_fold$f.json 2> $O/bench_err$f.txt
  python - <<PY
import json
d=json.load(open("$O/bench_ivit_fold$f.json"))
print("fold_wide=$f", d["value"], "img/s", d["ms_per_step"], "ms", {k: round(v,1) for k,v in d.get("in_step_us_per_launch",{}).items()})
r=d.get("rooflines",{})
print("   launches/step:", {k: r[k].get("launches_per_step") for k in ("norm","qk_norm") if k in r})
PY
done 2>&1 | tee $O/ivit_fold_ab.txt
