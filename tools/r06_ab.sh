#!/bin/bash
# GPU box: round 6's two build / policy A/Bs on one box -> gpurun_out/r06_nolicm_ab.txt, gpurun_out/r06_one_launch_ab.txt
#   (1) machine LICM on / off for the two parts of wf_resolve.hip (tools/build_alt_res.sh: lib_ship = the Makefile's choice,
#       lib_res4licm = the four-wave kernel WITH machine LICM, lib_res1nolicm = the one-wave kernel WITHOUT it)
#   (2) WF_RESOLVE_POLICY = four (shipped: the four-wave kernel for a flagged list of any length, one dispatch) / both (rounds 3-5:
#       both kernels enqueued, each reads the count on the device) -> gpurun_out/r06_one_launch_ab.txt (bench lines),
#       gpurun_out/r06_four_wave_always_ab.txt (tools/levels_ab.py)
#   (3) WF_RES4_PER_CU = 4 (shipped: as many as fit) / 3 / 2: residency of the four-wave float64 kernel -> gpurun_out/r06_res4_residency_ab.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_nolicm_ab.txt; : > $O
for l in ship res4licm res1nolicm; do
  echo "## build/alt/lib_$l.so" >> $O
  WFSTEP_LIB=build/alt/lib_$l.so WFSTEP_NO_AUTOBUILD=1 timeout 600 python tools/levels_ab.py 2>&1 | grep -v amdgpu.ids >> $O
done
O=gpurun_out/r06_one_launch_ab.txt; : > $O
for v in four both; do
  for cfg in cfg2 cfg3b cfg4; do
    echo "## WF_RESOLVE_POLICY=$v bench.py --config $cfg" >> $O
    WF_RESOLVE_POLICY=$v python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['headline_wind']
print('ms_per_step %.4f  kernel %.4f  float32-only %.4f -> strict %.4f (%d farms re-solved)' % (d['ms_per_step'], d['roofline']['kernel_ms'], e['float32_only']['ms_per_step'], e['with_float64_resolve']['ms_per_step'], e['n_resolved']))" >> $O
  done
done
O=gpurun_out/r06_four_wave_always_ab.txt; : > $O
for v in four both; do
  echo "## WF_RESOLVE_POLICY=$v" >> $O
  WF_RESOLVE_POLICY=$v timeout 600 python tools/levels_ab.py 2>&1 | grep -v amdgpu.ids >> $O
done
O=gpurun_out/r06_mode2_ab.txt; : > $O
for v in four both; do
  echo "## WF_RESOLVE_POLICY=$v (mode 2, every farm in float64: the four-wave kernel with levels for farms of 16 turbines and more / the one-wave kernel beyond a residency)" >> $O
  WF_RESOLVE_POLICY=$v python tools/mode2_timing.py 2>&1 | grep -v amdgpu.ids >> $O
done
O=gpurun_out/r06_res4_residency_ab.txt; : > $O
for pc in 4 3 2; do
  echo "## WF_RES4_PER_CU=$pc (blocks of the four-wave float64 kernel per CU at most; WF_RESOLVE_POLICY=both: lists beyond that residency go to the one-wave kernel, as in rounds 3-5 at 2)" >> $O
  WF_RESOLVE_POLICY=both WF_RES4_PER_CU=$pc timeout 600 python tools/levels_ab.py 2>&1 | grep -v amdgpu.ids | grep "HornsRev2 x 16384\|cfg4\|cfg5" >> $O
done
cat gpurun_out/r06_one_launch_ab.txt
