#!/bin/bash
# GPU box: the guard-band study of round 4 (tests/tools/band_study.py) repeated at twice the size on round 6's kernels, the shared-wind
# random-walk regime of the headline added -> gpurun_out/r06_band_study.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_band_study.txt; : > $O
run() { timeout 900 python tests/tools/band_study.py "$@" 2>&1 | grep -v amdgpu.ids >> $O; }
run HornsRev1_ 12 65536 reset
run HornsRev1_ 8 65536 wide
run HornsRev1_ 8 65536 shared
run HornsRev2_ 8 65536 shared
run HornsRev2_ 6 65536 reset
run Turb_TCRWP_ 8 65536 reset
run Ablaincourt_ 8 65536 wide
run fuzz 96 8192 wide
run fuzz 64 8192 reset
cat $O | cut -c1-160
