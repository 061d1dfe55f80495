mkdir -p gpurun_out/b4
python tools/attn2_skeleton.py > gpurun_out/b4/attn2_skeleton.txt 2>&1
cat gpurun_out/b4/attn2_skeleton.txt
