cd $GRAFT_REPO_ROOT
(BP=4096 TIME=0 timeout 900 python3 tests/tools/gpu_check.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_parity_stats.txt; tail -20 gpurun_out/r04_parity_stats.txt | cut -c1-300
timeout 900 python3 tests/tools/stress.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_stress.txt; cat gpurun_out/r04_stress.txt | cut -c1-250
timeout 600 python3 tests/tools/flag_stats.py 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_flag_stats.txt; grep -E "B=4096" gpurun_out/r04_flag_stats.txt | cut -c1-200
