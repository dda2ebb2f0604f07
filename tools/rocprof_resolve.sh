#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of tools/resolve_cost.py (HornsRev1 x 65536 and x 16384): average duration of the
# float32 step kernels, the compaction and the two float64 kernels -> gpurun_out/r03_v28_kernel_stats_resolve_*.csv
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for B in 65536 16384; do
  O=$R/gpurun_out/rp_resolve_$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/resolve_cost.py HornsRev1_ $B > $O.log 2> $O.err
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/summarize_rocprof.py $f $R/gpurun_out/r03_v28_kernel_stats_resolve_B$B.csv | head -8
  rm -rf $O
done
