set -e
cd wfcrl-env_amd/csrc
for occ in 2 3 4 5; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -DWF_RES_OCC=$occ -c -o wf_resolve.o wf_resolve.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libwfstep.so wf_kernels_1.o wf_kernels_2.o wf_kernels_ll.o wf_resolve.o wf_abi.o wf_model.o wf_dispatch.o wf_groups.o wf_wind_abi.o wf_env_abi.o
  echo "== WF_RES_OCC=$occ"
  (cd ../.. && python tools/resolve_cost.py HornsRev1_ 65536 2>&1 | tail -1)
done
