cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm/$n -- python $R/tools/gemm_vs_hipblaslt.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc_gemm/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemm256" not in k and "Cijk" not in k: continue
        # key by kernel name + grid size to separate shapes
        key=(k[:48], r.get("Grid_Size",""))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in sorted(agg.items()):
    print(k)
    print("   "+"  ".join("%s=%.4g"%(c, sum(v)/len(v)) for c,v in sorted(d.items())))
PY
