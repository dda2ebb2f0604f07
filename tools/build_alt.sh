#!/bin/bash
# A/B builds of the one-block kernel: tools/build_alt.sh NAME [extra hipcc flags, e.g. -DWF_X=1]  -> build/alt/lib_NAME.so
# (wf_kernels_ll.hip compiled with the flags and the Makefile's options, linked with the product's other objects;
#  time with tools/time_variants.py build/alt/lib_*.so on the GPU box.  build/ is not committed.)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
make -s -C wfcrl-env_amd/csrc > /dev/null
SCHED=${SCHED--mllvm -amdgpu-sched-strategy=iterative-ilp}  # SCHED="" for the default scheduler
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -fno-fast-math -ffp-contract=off -fno-slp-vectorize $SCHED"
mkdir -p build/alt
/opt/rocm/bin/hipcc $F "$@" -c -o build/alt/ll_$name.o wfcrl-env_amd/csrc/wf_kernels_ll.hip
C=wfcrl-env_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/alt/lib_$name.so $C/wf_kernels_1.o $C/wf_kernels_2.o build/alt/ll_$name.o $C/wf_resolve.o $C/wf_resolve4.o $C/wf_resolve4_mt.o \
  $C/wf_abi.o $C/wf_model.o $C/wf_dispatch.o $C/wf_groups.o $C/wf_wind_abi.o $C/wf_env_abi.o $C/wf_sort.o
echo built build/alt/lib_$name.so
