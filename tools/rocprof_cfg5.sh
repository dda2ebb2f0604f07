#!/bin/bash
# GPU box: kernel-time breakdown of BASELINE configs[4] (HornsRev2 x 131072, direction sweep re-set every step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cfg5prof
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --config cfg5 --no-cpu-baseline --no-env-leg > $O/bench_cfg5_under_rocprof.json 2> $O/rocprof.err
cd $R
f=$(find $O/rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f $R/gpurun_out/r03_v31_kernel_stats_bench_cfg5.csv
cp $O/bench_cfg5_under_rocprof.json $R/gpurun_out/r03_v31_bench_cfg5_under_rocprof.json
rm -rf $O
