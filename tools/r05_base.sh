cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_base
mkdir -p $O
for g in 16 8 4x2 2x2; do
  for b in 8192 65536; do
    WF_LL_G=$g B=$b timeout 300 python tools/ll_stamps.py build/alt/lib_stamp.so > $O/stamps_G${g}_B$b.txt 2>&1
  done
done
timeout 600 python tools/small_batch_sweep.py > $O/small_batch_sweep.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
cp wfcrl-env_amd/libwfstep.so /tmp/lib_keep.so
timeout 600 bash tools/res_stamps.sh > $O/res_stamps.txt 2>&1
