set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python tools/msda9_variants.py > $O/msda9_variants.txt 2>&1; grep -v amdgpu.ids $O/msda9_variants.txt
MSDA9_LIB=visionllm_amd/_build_abl/libmsda9_tw6.so timeout 900 python tools/gpu_passes/dbg_msda9_race.py 100 18 > $O/race_tw6.txt 2>&1; grep -v amdgpu.ids $O/race_tw6.txt | grep "runs differing\|level" | awk '{print $1,$2,$8,$13,$14,$15,$16}' | tail -30
