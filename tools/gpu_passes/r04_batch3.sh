set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python tools/gpu_passes/dbg_msda9_race.py 300 20 > $O/race9d.txt 2>&1; grep -v amdgpu.ids $O/race9d.txt | grep "mixed\|level" | tail -40
timeout 300 python tools/msda9_ab.py > $O/msda9_ab2.txt 2>&1; grep -v amdgpu.ids $O/msda9_ab2.txt
