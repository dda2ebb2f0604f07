#!/bin/bash
# GPU box: A/B builds of wf_resolve.hip: bash tools/res_ab.sh "-DWF_RES_OCC=2" "-DWF_RES_OCC=1 -DWF_RES_UNROLL_J=1" ...
cd $GRAFT_REPO_ROOT/wfcrl-env_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off -fno-slp-vectorize $v -c -o wf_resolve.o wf_resolve.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libwfstep.so wf_kernels_1.o wf_kernels_2.o wf_kernels_ll.o wf_resolve.o wf_abi.o wf_model.o wf_dispatch.o wf_groups.o wf_wind_abi.o wf_env_abi.o
  echo "== $v"
  (cd ../.. && python tools/resolve_cost.py HornsRev1_ 65536 2>&1 | tail -1)
done
