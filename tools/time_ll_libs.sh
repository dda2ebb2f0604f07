#!/bin/bash
# GPU box: alternative builds of the one-block kernel, G from WF_LL_G (default 8)
export WF_LL_G=${WF_LL_G:-8}
for lay in ${LAYOUTS:-HornsRev1_}; do
  echo "# $lay G=$WF_LL_G"; LAYOUT=$lay python tools/time_variants.py "$@" 2>&1 | grep ms/step
done
