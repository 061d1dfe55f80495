#!/bin/bash
# round 5, pass 35: the full -m gpu suite + smoke on the final tree (after the prune of the round-2 backward)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
make -C visionllm_amd/csrc -j16 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r05o/final_pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1 | tee -a gpurun_out/r05o/final_pytest.txt
