#!/bin/bash
# GPU box: everything profiles/ cites for one kernel version.  Usage: tools/record_profiles.sh <tag>
#   bench line (with cpu_baseline), the same command under rocprofv3 --kernel-trace --stats, PMC counters in
#   separate passes (tools/run_pmc.sh) for the headline kernel and for the on-the-fly kernel (bench.py --per-env-wind: a wind per farm), every BASELINE config,
#   parity statistics (tests/tools/gpu_check.py), wind / series modes, the two-rank bench line.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 2 --steps 20 --no-env-leg 2> $O/bench_n2.err | grep "^{" > $O/bench_n2_one_gpu_gloo.json
python3 tools/all_configs.py 40 > $O/all_configs.txt 2>&1
(BP=4096 TIME=0 python3 tests/tools/gpu_check.py) > $O/parity_stats.txt 2>&1
python3 tools/gs_sweep.py 2>&1 | grep ms/step > $O/variant_sweep.txt
(LLGS="0 4 8 16 4x2" LAYOUTS="HornsRev1_ HornsRev2_ Turb_TCRWP_" bash tools/time_ll.sh) > $O/one_block_kernel_sweep.txt 2>&1
python3 tools/time_wind_modes.py 2>&1 | grep ms/step > $O/wind_modes.txt
python3 tools/time_series_mode.py 64 2.0 2>&1 | grep ms/step > $O/series_modes.txt
python3 tools/latency_b1.py 2>&1 | grep update_command > $O/latency_b1.txt
python3 tests/tools/flag_stats.py 4096 > $O/flag_stats.txt 2>&1
(python3 tools/resolve_cost.py HornsRev1_ 65536; python3 tools/resolve_cost.py HornsRev2_ 131072) 2>&1 | grep -v amdgpu.ids > $O/resolve_cost.txt
python3 tools/batch_sweep_fine.py 2>&1 | grep -v amdgpu.ids > $O/batch_sweep_fine.txt
bash tools/pmc_resolve.sh 2>&1 | grep -A12 counter_collection > $O/pmc_resolve.txt
bash tools/run_pmc.sh $tag > /dev/null 2>&1
python3 tools/parse_pmc.py $tag wf_step_ll_kernel > $O/pmc_cfg4.json
rm -rf $R/gpurun_out/pmc_$tag
bash tools/run_pmc.sh ${tag}_fly --per-env-wind > /dev/null 2>&1
python3 tools/parse_pmc.py ${tag}_fly wf_step_ll_kernel > $O/pmc_cfg4_on_the_fly.json
rm -rf $R/gpurun_out/pmc_${tag}_fly
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --no-cpu-baseline --no-env-leg > $O/bench_under_rocprof.json 2> $O/rocprof.err
cd $R
f=$(find $O/rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f $O/kernel_stats_bench_cfg4.csv > /dev/null
rm -rf $O/rocprof
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --no-cpu-baseline --no-env-leg --per-env-wind > $O/bench_under_rocprof_on_the_fly.json 2> $O/rocprof_fly.err
cd $R
f=$(find $O/rocprof -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f $O/kernel_stats_bench_cfg4_on_the_fly.csv > /dev/null
rm -rf $O/rocprof
ls -la $O
