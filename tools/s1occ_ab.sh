cd $GRAFT_REPO_ROOT
O=gpurun_out/s1occ.txt; : > $O
for cfg in "16 8192 HornsRev1_" "8 16384 HornsRev1_" "8 16384 HornsRev2_" "4 24576 HornsRev1_" "8 12288 HornsRev1_" "16 6144 HornsRev2_"; do
  set -- $cfg
  echo "## WF_LL_G=$1 B=$2 $3" >> $O
  WF_CALIBRATE=0 WF_LL_G=$1 B=$2 LAYOUT=$3 timeout 200 python tools/time_variants.py wfcrl-env_amd/libwfstep.so build/alt/lib_s1occ2.so 2>&1 | grep -v amdgpu >> $O
done
cat $O
