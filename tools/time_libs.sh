#!/bin/bash
# GPU box: table path and on-the-fly path timings of alternative builds (build/alt/lib_*.so)
for lay in ${LAYOUTS:-HornsRev1_}; do
  echo "# table path $lay"; LAYOUT=$lay python tools/time_variants.py "$@" 2>&1 | grep ms/step
  echo "# on the fly $lay"; LAYOUT=$lay WF_NO_PAIR_TABLE=1 python tools/time_variants.py "$@" 2>&1 | grep ms/step
done
