#!/bin/bash
# GPU box: the handle's state machine under many more random sequences than the test suite's fixed sample
# (tests/tools/fuzz_api.py; round 4 found the stale launch order after a grouped launch this way).  -> gpurun_out/$1
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04_fuzz_api_long.txt}
: > $O
for seed in ${SEEDS:-8101 8102 8103 8104 8105 8106}; do
  echo "## python tests/tools/fuzz_api.py 60 50 $seed" >> $O
  timeout 500 python tests/tools/fuzz_api.py 60 50 $seed 2>&1 | grep -v amdgpu.ids | grep -E "^BAD|^   |^api" | cut -c1-1200 >> $O
done
echo "## python tests/tools/fuzz_env.py 100 8201" >> $O
timeout 600 python tests/tools/fuzz_env.py 100 8201 2>&1 | grep -E "^BAD|^env" | cut -c1-600 >> $O
grep -E "^##|violations" $O | cut -c1-200
