cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn/$n -- python $R/tools/attn_only.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(R+"/gpurun_out/pmc_attn/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(R+"/gpurun_out/pmc_attn/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("median duration us", sorted(dur)[len(dur)//2]/1e3)
for c,v in sorted(agg.items()): print("  %-28s %.4g"%(c, sum(v)/len(v)))
PY
