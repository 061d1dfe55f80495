set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b50
timeout 240 ./tools/probes/store_data_hazard > gpurun_out/b50/store_data_hazard.txt 2>&1; cat gpurun_out/b50/store_data_hazard.txt
