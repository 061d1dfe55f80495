cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -8 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2 | tee $O/smoke.txt
