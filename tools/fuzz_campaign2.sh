#!/bin/bash
# GPU box: a second fuzz campaign on the round's final build, other seeds -> gpurun_out/r03_fuzz_final2.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_fuzz_final2.txt
: > $O
for seed in 601 602; do python tests/tools/fuzz_parity.py 2000 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 >> $O; done
WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 2000 611 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 >> $O
WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py 1000 621 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 >> $O
WF_FUZZ_VEER=1 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 600 622 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400 >> $O
for seed in 631 632; do python tests/tools/fuzz_api.py 80 50 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600 >> $O; done
python tests/tools/fuzz_env.py 60 641 2>&1 | grep -v amdgpu.ids | tail -3 >> $O
grep -E "violations|BAD" $O
