cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest10.txt 2>&1; tail -4 gpurun_out/r04_pytest10.txt
python tools/small_batch_sweep.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_small_batch_sweep.txt; cat gpurun_out/r04_small_batch_sweep.txt
python tools/all_configs.py 40 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_all_configs.txt; cut -c1-200 gpurun_out/r04_all_configs.txt
python tools/latency_b1.py 2>&1 | grep update_command > gpurun_out/r04_latency_b1.txt; cat gpurun_out/r04_latency_b1.txt
