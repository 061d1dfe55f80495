set -x
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
O=gpurun_out/b51
mkdir -p $O; rm -rf $O/fetch $O/write
cp profiles/pmc_traffic.json $O/pmc_traffic.json; cp profiles/pmc_traffic_raw.json $O/pmc_traffic_raw.json 2>/dev/null
(cd /tmp; timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/fetch -- python $R/bench.py --workload internvit6b --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/write -- python $R/bench.py --workload internvit6b --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
python tools/collect_pmc.py $O/fetch $O/write $O/pmc_traffic.json internvit6b | head -24
find $O/fetch $O/write -name '*.csv' -size +1M -delete
