#!/bin/bash
# round 5, pass 25: race screen incl. the backward + all MSDA / DCNv3 tests on the rebuilt library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests/test_race_screen_gpu.py tests/test_msda_gpu.py tests/test_dcnv3_gpu.py -x -q 2>&1 | tail -8 | tee gpurun_out/r05o/msda_dcn_tests.txt
