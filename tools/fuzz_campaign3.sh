#!/bin/bash
# GPU box: third campaign on the build with the core-factor series, re-solve on (every farm strict) -> gpurun_out/r03_fuzz_final3.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_fuzz_final3.txt
: > $O
for seed in 811 812; do WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 2000 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O; done
WF_FUZZ_VEER=1 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 1200 821 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O
python tests/tools/fuzz_parity.py 1500 831 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O
python tests/tools/fuzz_api.py 60 50 841 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900 >> $O
grep -E "violations|BAD" $O | cut -c1-300
