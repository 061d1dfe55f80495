# MFMA-pipe utilisation of the persistent 8-phase GEMM against one workgroup per tile, by PMC (separate passes, kernel-trace only):
#   SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES (x4 SIMDs) and the bf16 MFMA op count, ViT-L qkv / fc1 shapes at M = 23080.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm_p
rm -rf $O; mkdir -p $O
for mode in persistent tilewise; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    for shape in "23080 3072 1024" "23080 4096 1024"; do
      s=$(echo $shape | tr ' ' 'x')
      if [ $mode = tilewise ]; then export VLLM_GEMM_PERSIST=0; else unset VLLM_GEMM_PERSIST; fi
      timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$mode/$s/$n -- python $R/tools/gemm_one.py $shape > /dev/null 2>&1
    done
  done
done
unset VLLM_GEMM_PERSIST
python - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
for mode in ("persistent", "tilewise"):
    for s in ("23080x3072x1024", "23080x4096x1024"):
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{R}/gpurun_out/pmc_gemm_p/{mode}/{s}/*/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "gemm256" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        m = {k: sum(v) / len(v) for k, v in agg.items()}
        if not m:
            print(mode, s, "no counters"); continue
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CU_CYCLES", 1), 1) / 4.0
        print(f"{mode:10s} {s}: MFMA busy / (CU busy x 4 SIMDs) = {busy:.3f}   " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(m.items())))
PY
