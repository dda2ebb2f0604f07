#!/bin/bash
# GPU box: per-phase stamps of the four-wave float64 kernel (sequential and level stages): a -DWF_RES_STAMP build of part 2
cd $GRAFT_REPO_ROOT/wfcrl-env_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off -fno-slp-vectorize -mllvm -disable-machine-licm -DWF_RES_STAMP -c -o /tmp/wf_resolve4_stamp.o wf_resolve4.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libwfstep.so wf_kernels_1.o wf_kernels_2.o wf_kernels_ll.o wf_resolve.o /tmp/wf_resolve4_stamp.o wf_resolve_mt.o wf_resolve4_mt.o wf_abi.o wf_model.o wf_dispatch.o wf_groups.o wf_wind_abi.o wf_env_abi.o wf_sort.o
cd ../.. && python tools/res4_stamps.py
