# which kernels does hipBLASLt pick for the ViT-L GEMM shapes (names carry the macro tile / pipeline parameters)
set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b10
rm -rf gpurun_out/b10/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b10/prof -o g -- python tools/gemm_vs_hipblaslt.py > gpurun_out/b10/gemm_vs_hipblaslt.txt 2>/dev/null
f=$(find gpurun_out/b10/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/b10/kernel_stats.csv
find gpurun_out/b10/prof -type f -size +1M -delete
cut -c1-400 gpurun_out/b10/kernel_stats.csv | head -30
