# round 5: PMC passes (counters only, --kernel-trace) of the two north-star kernels with this round's defaults
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
rm -rf gpurun_out/pmc_attn gpurun_out/pmc4
bash tools/pmc_attn_run.sh 2>&1 | grep -v amdgpu | tee $O/pmc_attn.txt
MSDA_MODES=1 bash tools/pmc_msda_run.sh 2>&1 | grep -v amdgpu | tee $O/pmc_msda.txt
rm -rf gpurun_out/pmc_attn gpurun_out/pmc4
