#!/bin/bash
# GPU box: what profiles/ cites for round 6's final build.  Usage: tools/record_profiles_r06.sh <tag>  -> gpurun_out/<tag>_*
tag=${1:-r06_final}
R=$GRAFT_REPO_ROOT; cd $R
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for cfg in cfg5 cfg3b cfg2; do python3 bench.py --config $cfg --steps 20 --cpu-seconds 4 > gpurun_out/${tag}_bench_$cfg.json 2> gpurun_out/${tag}_bench_$cfg.err; done
python3 bench.py --per-env-wind --steps 20 --no-cpu-baseline --no-env-leg 2>/dev/null | grep "^{" > gpurun_out/${tag}_bench_cfg4_per_env_wind.json
PMC_KERNEL="wf_step_ll_kernel<2, 2, true, true" bash tools/quick_pmc.sh ${tag} > /dev/null 2>&1
PMC_KERNEL="wf_step_ll_kernel<2, 2, false, false" bash tools/quick_pmc.sh ${tag}_fly --per-env-wind > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_rocprof -- python3 $R/bench.py --no-cpu-baseline --no-env-leg > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_rocprof.err
cd $R
f=$(find gpurun_out/${tag}_rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f gpurun_out/${tag}_kernel_stats_bench_cfg4.csv > /dev/null
rm -rf gpurun_out/${tag}_rocprof
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_rocprof -- python3 $R/bench.py --no-cpu-baseline --no-env-leg --per-env-wind > /dev/null 2> $R/gpurun_out/${tag}_rocprof_fly.err
cd $R
f=$(find gpurun_out/${tag}_rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f gpurun_out/${tag}_kernel_stats_bench_cfg4_on_the_fly.csv > /dev/null
rm -rf gpurun_out/${tag}_rocprof
(python3 tools/small_batch_sweep.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_small_batch_sweep.txt
(LAYOUT=HornsRev2_ python3 tools/small_batch_sweep.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_small_batch_sweep_hornsrev2.txt
python3 tools/latency_b1.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_latency_b1.txt
(python3 tools/levels_ab.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_levels_ab.txt
(python3 tests/tools/levels_check.py 96) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_levels_check.txt
(python3 tools/env_cost.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_env_cost.txt
ls -la gpurun_out/${tag}_*
