#!/bin/bash
# GPU box: round 6's two build / policy A/Bs on one box -> gpurun_out/r06_nolicm_ab.txt, gpurun_out/r06_one_launch_ab.txt
#   (1) machine LICM on / off for the two parts of wf_resolve.hip (tools/build_alt_res.sh: lib_ship = the Makefile's choice,
#       lib_res4licm = the four-wave kernel WITH machine LICM, lib_res1nolicm = the one-wave kernel WITHOUT it)
#   (2) WF_RESOLVE_ONE_LAUNCH = 1 (shipped) / 0: one or two float64 dispatches behind a step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_nolicm_ab.txt; : > $O
for l in ship res4licm res1nolicm; do
  echo "## build/alt/lib_$l.so" >> $O
  WFSTEP_LIB=build/alt/lib_$l.so WFSTEP_NO_AUTOBUILD=1 timeout 600 python tools/levels_ab.py 2>&1 | grep -v amdgpu.ids >> $O
done
O=gpurun_out/r06_one_launch_ab.txt; : > $O
for v in 1 0; do
  for cfg in cfg2 cfg3b cfg4; do
    echo "## WF_RESOLVE_ONE_LAUNCH=$v bench.py --config $cfg" >> $O
    WF_RESOLVE_ONE_LAUNCH=$v python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['headline_wind']
print('ms_per_step %.4f  kernel %.4f  float32-only %.4f -> strict %.4f (%d farms re-solved)' % (d['ms_per_step'], d['roofline']['kernel_ms'], e['float32_only']['ms_per_step'], e['with_float64_resolve']['ms_per_step'], e['n_resolved']))" >> $O
  done
done
cat gpurun_out/r06_one_launch_ab.txt
