set -x
mkdir -p gpurun_out/b2
python tools/attn2_ablate.py 0 > gpurun_out/b2/attn2_ablate.txt 2>&1
cat gpurun_out/b2/attn2_ablate.txt
