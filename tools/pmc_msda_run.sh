cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc4/$n -- python $R/tools/pmc_msda.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc4/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "msda" not in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()):
        print("   %-24s %.4g (n=%d)"%(c, sum(v)/len(v), len(v)))
PY
