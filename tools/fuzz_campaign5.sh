#!/bin/bash
# GPU box: round-4 campaign on the build that carries the far-pair skip -> gpurun_out/$1 (default r04_fuzz_head.txt)
# legs: plain x2, re-solve on, veer, GCH internals, the far-pair-skip leg (dense layouts, randomised wake models, the
# table-path one-block families forced) with and without the re-solve, API sessions, env episodes
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04_fuzz_head.txt}
: > $O
leg() { echo "## $*" >> $O; env "$@" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-900 >> $O; }
for seed in 1011 1012; do leg python tests/tools/fuzz_parity.py 1500 $seed; done
leg WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 1500 1021
leg WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py 800 1031
leg WF_FUZZ_GCH=1 python tests/tools/fuzz_parity.py 800 1036
for seed in 1041 1042; do leg WF_FUZZ_SKIP=1 python tests/tools/fuzz_parity.py 1500 $seed; done
leg WF_FUZZ_SKIP=1 FUZZ_WS=2.5,26 python tests/tools/fuzz_parity.py 1000 1043
leg WF_FUZZ_SKIP=1 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 1000 1044
for seed in 1051 1052; do leg python tests/tools/fuzz_api.py 60 50 $seed; done
leg python tests/tools/fuzz_env.py 80 1061
grep -E "violations|BAD" $O | cut -c1-300
