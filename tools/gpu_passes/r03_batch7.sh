mkdir -p gpurun_out/b7
for sg in 0 600 1100 1700 2500 4000; do
  echo "== VLLM_ATTN_STAGGER=$sg"
  VLLM_ATTN_STAGGER=$sg python tools/bench_attn.py 64 2>&1 | grep -v amdgpu
done > gpurun_out/b7/attn_stagger.txt 2>&1
cat gpurun_out/b7/attn_stagger.txt
