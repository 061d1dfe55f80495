set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
mkdir -p gpurun_out/b14
timeout 600 python -m pytest tests/test_msda_gpu.py -m gpu -q -x -k "backward or bwd or grad" > gpurun_out/b14/pytest_bwd.txt 2>&1; echo "rc $?" >> gpurun_out/b14/pytest_bwd.txt
tail -12 gpurun_out/b14/pytest_bwd.txt
timeout 300 python tools/msda_bwd_phases.py libmsdabwd_mfma_prof.so > gpurun_out/b14/bwd_mfma_phases.txt 2>&1
cat gpurun_out/b14/bwd_mfma_phases.txt
