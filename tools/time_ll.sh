#!/bin/bash
# GPU box: the one-block-at-a-time kernel at each lane-group width against the register-slot kernel (table path)
for lay in ${LAYOUTS:-HornsRev1_}; do
  echo "# $lay"
  for g in ${LLGS:-0 4 8 16 4x2 8x2}; do
    if [ $g = 0 ]; then export WF_LL=0; unset WF_LL_G; else unset WF_LL; export WF_LL_G=$g; fi
    echo -n "LL_G=$g  "; LAYOUT=$lay python tools/time_variants.py wfcrl-env_amd/libwfstep.so 2>&1 | grep ms/step | head -1
  done
done
