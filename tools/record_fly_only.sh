tag=${1:-r02_fly}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
cd $R
bash tools/run_pmc.sh ${tag}_fly --per-env-wind > /dev/null 2>&1
python3 tools/parse_pmc.py ${tag}_fly wf_step_ll_kernel > $O/pmc_cfg4_on_the_fly.json
rm -rf $R/gpurun_out/pmc_${tag}_fly
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --no-cpu-baseline --no-env-leg --per-env-wind > $O/bench_under_rocprof_on_the_fly.json 2> $O/rocprof_fly.err
cd $R
f=$(find $O/rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f $O/kernel_stats_bench_cfg4_on_the_fly.csv > /dev/null
rm -rf $O/rocprof
head -4 $O/kernel_stats_bench_cfg4_on_the_fly.csv; cat $O/pmc_cfg4_on_the_fly.json | head -c 600
