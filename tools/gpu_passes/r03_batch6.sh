mkdir -p gpurun_out/b6
python tools/msda7_ablate.py encoder_like > gpurun_out/b6/msda7_ablate.txt 2>&1
cat gpurun_out/b6/msda7_ablate.txt
