# One GPU-box pass that regenerates what profiles/r06_* quote at the end of round 6: full -m gpu suite, smoke, the default bench
# line (ViT-L + the internvit6b key), rocprofv3 kernel stats of both workloads (csv), FETCH_SIZE / WRITE_SIZE PMC passes (separate
# runs, kernel-trace only), the MSDA backward phase clocks.  The individual passes of the round, as they were run, are in
# tools/gpu_passes/ (round 2's version of this script: git history).
#   gpurun --timeout 2700 -- 'bash tools/run_gpu_round.sh'      then copy gpurun_out/r06z/* into profiles/ under their r06_ names
set -x
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
make -C visionllm_amd/csrc -j16 2>&1 | tail -1   # (a stale .so once produced a wrong figure: rebuild whatever is out of date)
O=gpurun_out/r06z
mkdir -p $O
rm -f gpurun_out/parity_contract.jsonl gpurun_out/ulp_table.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
cp gpurun_out/parity_contract.jsonl $O/parity_contract.jsonl 2>/dev/null; cp gpurun_out/ulp_table.jsonl $O/ulp_table.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; tail -c 300 $O/bench_line.json; cp gpurun_out/bench_detail.json $O/bench_detail.json
rm -rf $O/prof_v $O/prof_i $O/fetch $O/write
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v -o bench -- python bench.py --workload vitl --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_prof_line_vitl.json 2> /dev/null
f=$(find $O/prof_v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > $O/bench_kernel_stats_vitl.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_i -o bench -- python bench.py --workload internvit6b --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_prof_line_internvit6b.json 2> /dev/null
f=$(find $O/prof_i -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > $O/bench_kernel_stats_internvit6b.csv
find $O -name '*kernel_trace*' -delete
find $O/prof_v $O/prof_i -type f -size +1M -delete
(cd /tmp; timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/fetch -- python $R/bench.py --workload vitl --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/write -- python $R/bench.py --workload vitl --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/collect_pmc.py $O/fetch $O/write $O/pmc_traffic.json vitl | head -30   # copy to profiles/pmc_traffic.json: bench.py reads `traffic` from there
find $O/fetch $O/write -name '*.csv' -size +2M -delete
timeout 300 python tools/prof_msda9.py 2>&1 | grep -v amdgpu | tee $O/msda9_phases.txt
# InternViT-6B traffic after the banded tile order (fc1 traffic <= 3x algorithmic)
(cd /tmp; timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/fetch_i -- python $R/bench.py --workload internvit6b --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
 timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/write_i -- python $R/bench.py --workload internvit6b --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/collect_pmc.py $O/fetch_i $O/write_i $O/pmc_traffic_internvit6b.json internvit6b | head -30
find $O/fetch_i $O/write_i -name '*.csv' -size +2M -delete
