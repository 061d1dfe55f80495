set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
for sk in 1 0; do
rm -rf $O/prof$sk
VLLM_GEMM_SKINNY=$sk timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$sk -o layer -- python tools/bench_msda_layer.py --case encoder --only fused > $O/line$sk.txt 2>/dev/null
f=$(find $O/prof$sk -name '*kernel_stats.csv' | head -1); head -6 "$f" | cut -c1-150 > $O/layer_kernel_stats_skinny$sk.csv; cat $O/layer_kernel_stats_skinny$sk.csv; cat $O/line$sk.txt
python - <<PY
import csv,glob
f=glob.glob("$O/prof$sk/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if 'vllm' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 8 kernels = two layer calls: print start gaps
prev=None
for r in rows[-8:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(r['Kernel_Name'][:60].ljust(60), 'dur %.1f us'%((e-s)/1e3), 'gap %.1f us'%(((s-prev)/1e3) if prev else 0))
    prev=e
PY
find $O/prof$sk -type f -size +1M -delete
done
