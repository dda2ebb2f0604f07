#!/bin/bash
# A/B builds of the float64 re-solve: tools/build_alt_res.sh NAME [extra hipcc flags, e.g. -DRES_LV_SCHED_LIMIT=0]  -> build/alt/lib_NAME.so
# (both parts of wf_resolve.hip compiled with the flags — part 2, the four-wave kernel, without machine LICM as in the
# Makefile — and linked with the product's other objects)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C wfcrl-env_amd/csrc > /dev/null
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -fno-fast-math -ffp-contract=off -fno-slp-vectorize"
mkdir -p build/alt
C=wfcrl-env_amd/csrc
/opt/rocm/bin/hipcc $F "$@" -c -o build/alt/res_$name.o $C/wf_resolve.hip
/opt/rocm/bin/hipcc $F ${RES4_LICM:--mllvm -disable-machine-licm} "$@" -c -o build/alt/res4_$name.o $C/wf_resolve4.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/alt/lib_$name.so $C/wf_kernels_1.o $C/wf_kernels_2.o $C/wf_kernels_ll.o build/alt/res_$name.o build/alt/res4_$name.o \
  $C/wf_resolve_mt.o $C/wf_resolve4_mt.o $C/wf_abi.o $C/wf_model.o $C/wf_dispatch.o $C/wf_groups.o $C/wf_wind_abi.o $C/wf_env_abi.o $C/wf_sort.o
echo built build/alt/lib_$name.so
