set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O; rm -f $O/ab_libs.txt $O/bench_ab.txt
timeout 900 python -m pytest tests/test_vit_gpu.py -q -x -k "half_height or persistent or gemm" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for i in 1 2 3; do
VLLM_HIP_LIB=$PWD/visionllm_amd/_build_old/libvllm_hip.so python tools/ab_libs.py 2>&1 | tail -1 >> $O/ab_libs.txt
python tools/ab_libs.py 2>/dev/null | tail -1 >> $O/ab_libs.txt
done
cat $O/ab_libs.txt
for i in 1 2 3; do
VLLM_GEMM_HALF_TAIL=0 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('half_tail=0', r['value'], r['ms_per_step'], r['clocks']['sclk_mhz_median'], r['rooflines']['gemm_qkv']['us_per_launch'])" >> $O/bench_ab.txt
VLLM_GEMM_HALF_TAIL=1 python bench.py --workload vitl --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('half_tail=1', r['value'], r['ms_per_step'], r['clocks']['sclk_mhz_median'], r['rooflines']['gemm_qkv']['us_per_launch'])" >> $O/bench_ab.txt
done
cat $O/bench_ab.txt
