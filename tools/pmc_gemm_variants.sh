# PMC comparison of the GEMM schedules on one shape (default 4096^3): ours variant 2 (16x16x32), 4 (32x32x16), hipBLASLt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SHAPE=${SHAPE:-"4096 4096 4096"}
rm -rf $R/gpurun_out/pmc_gv
for mode in v2 v4 lib; do
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_CMD_FIFO_FULL"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-30)
    if [ $mode = lib ]; then arg=lib; var=0; else arg=ours; var=${mode#v}; fi
    VLLM_GEMM_VARIANT=$var timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gv/$mode/$n -- python $R/tools/gemm_one.py $SHAPE $arg > /dev/null 2>&1
  done
done
python - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for mode in ("v2","v4","lib"):
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob(R+f"/gpurun_out/pmc_gv/{mode}/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "gemm256" in k or "Cijk" in k: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(R+f"/gpurun_out/pmc_gv/{mode}/*/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "gemm256" in k or "Cijk" in k: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    print(mode, "median us", sorted(dur)[len(dur)//2]/1e3 if dur else None)
    for c,v in sorted(agg.items()): print("   %-28s %.4g"%(c, sum(v)/len(v)))
PY
