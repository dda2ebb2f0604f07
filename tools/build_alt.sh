#!/bin/bash
# usage: tools/build_alt.sh NAME [extra hipcc flags...]   -> build/alt/lib_NAME.so  (single translation unit, WF_KSET=0)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
F="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -fno-fast-math -ffp-contract=off -fno-slp-vectorize"
mkdir -p build/alt
/opt/rocm/bin/hipcc $F -DWF_KSET=0 "$@" -c -o build/alt/k_$name.o wfcrl-env_amd/csrc/wf_kernels.hip
[ -f build/alt/abi.o ] && [ build/alt/abi.o -nt wfcrl-env_amd/csrc/wf_abi.hip ] || /opt/rocm/bin/hipcc $F -c -o build/alt/abi.o wfcrl-env_amd/csrc/wf_abi.hip
/opt/rocm/bin/hipcc $F $LLFLAGS -c -o build/alt/ll_$name.o wfcrl-env_amd/csrc/wf_kernels_ll.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/alt/lib_$name.so build/alt/k_$name.o build/alt/ll_$name.o build/alt/abi.o
echo built build/alt/lib_$name.so
