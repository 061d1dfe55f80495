# round 2, GPU pass O: InternViT-6B full-depth drift; kernel stats of the fused MSDA layer
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_ivit_depth_gpu.py -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee gpurun_out/r02o_ivit_drift_test.txt
head -3 gpurun_out/ivit_drift.jsonl; tail -2 gpurun_out/ivit_drift.jsonl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02o_prof -o layer -- python tools/bench_msda_layer.py > /dev/null 2> gpurun_out/r02o_prof_err.txt
f=$(find gpurun_out/r02o_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -30 "$f" | tee gpurun_out/r02o_layer_kernel_stats.csv
find gpurun_out/r02o_prof -name '*kernel_trace*' -delete
