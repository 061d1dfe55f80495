cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_INSTS_LDS[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAVES_[A-Z_]*\|SQ_LEVEL_WAVES\|SQ_OCC[A-Z_]*\|SQ_WAIT_INST_[A-Z_]*" | sort -u | tr '\n' ' '
echo
rm -rf $R/gpurun_out/pmc_attn2
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LEVEL_WAVES" "SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAVES"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn2/$n -- python $R/tools/attn_only.py > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(list); dur=[]
for f in glob.glob(R+"/gpurun_out/pmc_attn2/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(R+"/gpurun_out/pmc_attn2/*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "attn_fwd" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("median duration us", sorted(dur)[len(dur)//2]/1e3 if dur else None)
for c,v in sorted(agg.items()): print("  %-28s %.4g"%(c, sum(v)/len(v)))
PY
