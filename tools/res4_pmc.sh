#!/bin/bash
# GPU box: SQ counters of the four-wave float64 kernel on the HornsRev2 x 16384 share (39 flagged farms, one per CU), with helper
# waves on every launch / on none -> gpurun_out/r06_res4_pmc.txt.  (rocprofv3 --pmc with --kernel-trace only; the program itself after --)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r06_res4_pmc.txt; : > $O
for h in 2 0; do
  export WF_RES4_HELPERS=$h
  rm -rf $R/gpurun_out/res4pmc
  timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/res4pmc -- python3 $R/tools/res4_hr2_share.py 20 > /dev/null 2>&1
  echo "## WF_RES4_HELPERS=$h (2: 512-thread launches, eight waves per farm; 0: 256 threads, four waves): wf_resolve4_kernel, per launch" >> $O
  python3 - >> $O <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/res4pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if r["Kernel_Name"].startswith("wf_resolve4_kernel"):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"  {k:22s} {acc[k] / n[k]:16.0f}   ({n[k]} launches)")
kt = glob.glob("$R/gpurun_out/res4pmc/**/*kernel_trace.csv", recursive=True)
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if r["Kernel_Name"].startswith("wf_resolve4_kernel")]
dur = sorted(dur)[len(dur) // 2]
FARMS = 39  # flagged on this workload (tools/res4_hr2_share.py prints it): one block each, on a CU of its own
act = acc["SQ_ACTIVE_INST_VALU"] / n["SQ_ACTIVE_INST_VALU"] * 4  # (the counter is in units of four cycles)
ins = acc["SQ_INSTS_VALU"] / n["SQ_INSTS_VALU"]
print(f"  median duration under the counters {dur / 1e3:.1f} us = {dur * 2.4:.0f} cycles at 2.4 GHz; per farm {ins / FARMS:.0f} VALU wave-instructions;")
print(f"  VALU pipe busy on the {FARMS} CUs that hold a farm: {act / (FARMS * 4 * dur * 2.4):.2f} of their SIMD-cycles; {act / ins:.1f} busy cycles per instruction")
PY
  rm -rf $R/gpurun_out/res4pmc
done
cat $O
