#!/bin/bash
# GPU box: everything profiles/ cites for one kernel version.  Usage: tools/record_profiles.sh <tag>
#   bench line (with cpu_baseline), the same command under rocprofv3 --kernel-trace --stats, PMC counters in
#   separate passes (tools/run_pmc.sh), every BASELINE config, parity statistics (tests/tools/gpu_check.py).
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 tools/all_configs.py 40 > $O/all_configs.txt 2>&1
python3 tests/tools/gpu_check.py > $O/parity_stats.txt 2>&1
python3 tools/gs_sweep.py 2>&1 | grep ms/step > $O/variant_sweep.txt
python3 tools/time_wind_modes.py 2>&1 | grep ms/step > $O/wind_modes.txt
python3 tools/latency_b1.py 2>&1 | grep update_command > $O/latency_b1.txt
(for sd in 21 22; do python3 tests/tools/fuzz_parity.py 1000 $sd 2>&1 | grep -E "^BAD|^fuzz"; done; python3 tests/tools/fuzz_api.py 60 50 21 2>&1 | grep -E "^BAD|^api fuzz") > $O/fuzz.txt
bash tools/run_pmc.sh $tag > /dev/null 2>&1
python3 tools/parse_pmc.py $tag > $O/pmc_cfg4.json
rm -rf $R/gpurun_out/pmc_$tag
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.err
cd $R
f=$(find $O/rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f $O/kernel_stats_bench_cfg4.csv > /dev/null
rm -rf $O/rocprof
ls -la $O
