cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_layouts_gpu.py -m gpu -q -x > gpurun_out/r04_pytest12a.txt 2>&1; tail -25 gpurun_out/r04_pytest12a.txt | cut -c1-220
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_layouts_gpu.py > gpurun_out/r04_pytest12.txt 2>&1; tail -4 gpurun_out/r04_pytest12.txt
