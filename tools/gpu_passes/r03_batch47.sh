set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b47
bash tools/pmc_gemm_persistent.sh > gpurun_out/b47/pmc.txt 2>&1; tail -6 gpurun_out/b47/pmc.txt
rm -rf gpurun_out/pmc_gemm_p
