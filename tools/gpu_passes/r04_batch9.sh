set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 900 python -m pytest tests/test_msda_gpu.py -q -x -k "skinny or fused_layer or module" > $O/pytest_skinny.txt 2>&1; tail -5 $O/pytest_skinny.txt
rm -rf $O/prof1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o layer -- python tools/bench_msda_layer.py --case encoder --only fused > $O/line1.txt 2>/dev/null
f=$(find $O/prof1 -name '*kernel_stats.csv' | head -1); head -6 "$f" | cut -c1-150 > $O/layer_kernel_stats.csv; cat $O/layer_kernel_stats.csv; cat $O/line1.txt
find $O/prof1 -type f -size +1M -delete
