set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b49
for v in 2 6 10 14 2; do
VLLM_ATTN_VARIANT=$v timeout 300 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('attn variant $v:', round(d['value'], 1), 'img/s', round(d['ms_per_step'], 3), 'ms; attn', round(d['in_step_us_per_launch']['attn'], 1), 'us')" | tee -a gpurun_out/b49/attn_variants.txt
done
