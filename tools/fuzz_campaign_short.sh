#!/bin/bash
# GPU box: a short campaign after a change that should not move results (every leg of tools/fuzz_campaign5.sh at a third of its size)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04_fuzz_short.txt}
: > $O
leg() { echo "## $*" >> $O; timeout 900 env "$@" 2>&1 | grep -v amdgpu.ids | grep -E "^BAD|^fuzz|^api|^env" | cut -c1-1500 >> $O; }
leg python tests/tools/fuzz_parity.py 800 2011
leg WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 600 2021
leg WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py 300 2031
leg WF_FUZZ_SKIP=1 python tests/tools/fuzz_parity.py 600 2041
leg WF_FUZZ_SKIP=1 FUZZ_WS=2.5,26 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 400 2043
leg python tests/tools/fuzz_api.py 40 50 2051
leg python tests/tools/fuzz_env.py 40 2061
grep -E "^##|violations" $O | cut -c1-200
