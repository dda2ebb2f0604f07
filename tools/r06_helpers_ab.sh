#!/bin/bash
# GPU box: the helper waves of the four-wave float64 kernel (csrc/wf_resolve.hip: HELPER WAVES) -> gpurun_out/r06_helpers_ab.txt
#   WF_RES4_HELPERS = 0 (256-thread launches only: four waves per farm), 1 (shipped: 512 threads where the previous launch found a
#   short list), 2 (512 threads on every launch: a long list then runs two wide blocks per CU instead of three or four narrow ones)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_helpers_ab.txt; : > $O
for h in 0 1 2; do
  echo "## WF_RES4_HELPERS=$h  tools/levels_ab.py" >> $O
  WF_RES4_HELPERS=$h timeout 600 python tools/levels_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O
done
for h in 0 1; do
  echo "## WF_RES4_HELPERS=$h  tools/mode2_timing.py (every farm in float64)" >> $O
  WF_RES4_HELPERS=$h timeout 300 python tools/mode2_timing.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O | cut -c1-150
