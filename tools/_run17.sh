cd $GRAFT_REPO_ROOT
(timeout 300 python tests/tools/flag_stickiness.py HornsRev1_ 65536 12; timeout 300 python tests/tools/flag_stickiness.py HornsRev2_ 65536 8) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_flag_stickiness.txt; cat gpurun_out/r04_flag_stickiness.txt
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k far_skip 2>&1 | grep -E "far skip on|passed|failed" > gpurun_out/r04_far_skip_identity.txt; cat gpurun_out/r04_far_skip_identity.txt | cut -c1-200
