cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_vit_gpu.py -q -x -k "bridge or pixel_shuffle or norms" 2>&1 | tail -4 | tee $O/pytest_bridge.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x 2>&1 | tail -3 | tee $O/pytest_fullsize.txt
for f in 1 0; do VLLM_PS_FOLD=$f timeout 600 python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ps_fold=$f', round(d['ms_per_step'],2), 'ms', {k: round(v,1) for k,v in d['in_step_us_per_launch'].items() if 'bridge' in k})"; done | tee $O/ps_fold_ab.txt
