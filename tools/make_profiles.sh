#!/bin/bash
# usage: tools/make_profiles.sh <tag>   (on the GPU box; results under gpurun_out/profiles_<tag>/)
# kernel-trace stats of the default bench command, FETCH_SIZE / WRITE_SIZE passes, bench JSON lines
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the profiled command is the default bench command without its extra legs (CPU baseline, PCIe round trip, the
# fused-filter variant): only the timed sweep itself is under the profiler
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o out -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-fused-line > $O/${tag}_sweep_bench.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/${tag}_sweep_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c -d $O/pmc_$c -o out -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e --no-fused-line > $O/pmc_$c.log 2>&1
done
cd $R
python3 tools/pmc_fetch_write.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/${tag}_sweep_pmc_summary.json
# the summary has to be in profiles/ for bench.py to report roofline.traffic from it
cp $O/${tag}_sweep_pmc_summary.json $R/profiles/${tag}_sweep_pmc_summary.json
timeout 600 python3 bench.py --steps 10 --warmup 2 | tail -1 > $O/${tag}_bench_cfg3_sweep.json
timeout 300 python3 bench.py --workload cfg3 --steps 10 --warmup 2 --no-e2e | tail -1 > $O/${tag}_bench_cfg3_build.json
timeout 300 python3 bench.py --workload cfg2 --steps 20 --warmup 2 --no-e2e | tail -1 > $O/${tag}_bench_cfg2.json
timeout 600 python3 bench.py --workload cfg4 --steps 2 --warmup 1 | tail -1 > $O/${tag}_bench_cfg4.json
timeout 300 python3 bench.py --steps 10 --warmup 2 --force-merge --no-cpu-baseline --no-e2e | tail -1 > $O/${tag}_bench_cfg3_sweep_merged_n1.json
rm -rf $O/ks $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
