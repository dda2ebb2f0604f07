#!/bin/bash
# GPU box: the far-pair-skip leg of the layout fuzzer with the skip ON and OFF (WF_LL_FAR_SKIP=0 seeds
# wf_kernel_choice::far_skip at wf_create) on the same seeds: a violation that is the skip's shows up in one column only.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04_fuzz_skip_ab.txt}
: > $O
for seed in 1041 1044; do
  for fs in 1 0; do
    echo "## seed $seed WF_LL_FAR_SKIP=$fs" >> $O
    WF_LL_FAR_SKIP=$fs WF_FUZZ_SKIP=1 WF_FUZZ_RESOLVE=$([ $seed = 1044 ] && echo 1) python tests/tools/fuzz_parity.py ${2:-1000} $seed 2>&1 | grep -v amdgpu.ids | grep -E "^BAD|^fuzz" | cut -c1-1800 >> $O
  done
done
grep -E "^##|^fuzz" $O
