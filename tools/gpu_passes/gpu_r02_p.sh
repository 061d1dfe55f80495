# round 2, GPU pass P: per-kernel times of the fused MSDA layer (encoder shape), one launch structure per rocprofv3 run
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_msda_gpu.py -m gpu -x -q -k "empty_level or layer or f32 or prep or gemm" 2>&1 | tail -4
for s in fused; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02p_prof_$s -o layer -- python tools/bench_msda_layer.py --case encoder --only $s > gpurun_out/r02p_$s.txt 2> /dev/null
f=$(find gpurun_out/r02p_prof_$s -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee gpurun_out/r02p_layer_kernels_$s.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-100s calls %5s avg_us %9.1f  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find gpurun_out/r02p_prof_$s -name '*kernel_trace*' -delete
done
cat gpurun_out/r02p_fused.txt; timeout 300 python tools/bench_msda_layer.py | tail -2
