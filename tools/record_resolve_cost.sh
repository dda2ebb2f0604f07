cd $GRAFT_REPO_ROOT
(for B in 4096 16384 24576 32768 65536; do python tools/resolve_cost.py HornsRev1_ $B 2>&1 | grep -v amdgpu.ids | tail -2; done; python tools/resolve_cost.py HornsRev2_ 131072 2>&1 | grep -v amdgpu.ids; python tools/veer_rate.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r03_v28_resolve_cost.txt
cat gpurun_out/r03_v28_resolve_cost.txt
