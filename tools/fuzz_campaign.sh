#!/bin/bash
# GPU box: the round's fuzz campaign on the final build -> gpurun_out/r03_fuzz_final.txt
#   layout fuzzer (tests/tools/fuzz_parity.py) plain and with the float64 re-solve on; API-sequence fuzzer
#   (tests/tools/fuzz_api.py: sessions also toggle the re-solve, change the handle's kernel choice and set several
#   layouts per batch); the layout fuzzer with every case a wind-veer model; env fuzzer.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_fuzz_final.txt
: > $O
for seed in 301 302; do python tests/tools/fuzz_parity.py 1500 $seed 2>&1 | grep -v amdgpu.ids | tail -4 >> $O; done
for seed in 311 312; do WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 1500 $seed 2>&1 | grep -v amdgpu.ids | tail -4 >> $O; done
for seed in 321 322; do python tests/tools/fuzz_api.py 80 50 $seed 2>&1 | grep -v amdgpu.ids | tail -4 >> $O; done
WF_FUZZ_VEER=1 python tests/tools/fuzz_parity.py 800 341 2>&1 | grep -v amdgpu.ids | tail -3 >> $O
python tests/tools/fuzz_env.py 40 331 2>&1 | grep -v amdgpu.ids | tail -3 >> $O
cat $O
