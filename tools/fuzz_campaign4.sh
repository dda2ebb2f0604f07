#!/bin/bash
# GPU box: fourth campaign (last build of the round) -> gpurun_out/r03_fuzz_final4.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_fuzz_final4.txt
: > $O
for seed in 911 912; do python tests/tools/fuzz_parity.py 2000 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O; done
WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 2000 921 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O
WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py 1200 931 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O
for seed in 941 942; do python tests/tools/fuzz_api.py 60 50 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O; done
python tests/tools/fuzz_env.py 80 951 2>&1 | grep -v amdgpu.ids | tail -3 >> $O
grep -E "violations|BAD" $O | cut -c1-300
