cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_resolve_gpu.py -m gpu -q -x -k "far_skip or veer or one_block" > gpurun_out/r04_pytest15.txt 2>&1; tail -4 gpurun_out/r04_pytest15.txt
O=gpurun_out/r04_fuzz_veer2.txt; : > $O
for fs in 1 0; do echo "## WF_FUZZ_VEER=1 WF_LL_FAR_SKIP=$fs" >> $O; WF_LL_FAR_SKIP=$fs WF_FUZZ_VEER=1 timeout 900 python tests/tools/fuzz_parity.py 600 3031 2>&1 | grep -E "^BAD|^fuzz" | cut -c1-1500 >> $O; done; grep -E "^##|^fuzz" $O
timeout 300 python tools/veer_rate.py 2>&1 | grep -v amdgpu.ids | tail -12
