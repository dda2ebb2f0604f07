#!/bin/bash
# GPU box: counters of the float64 re-solve kernel (tools/resolve_cost.py workload), one group per rocprofv3 pass.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_resolve
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/tools/resolve_cost.py HornsRev1_ 65536 > $O/$name.log 2> $O/$name.err; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "resolve" in k:
            print(f, k)
            for cn, v in acc[k].items():
                print(f"   {cn}: {v / n[(k, cn)]:.4g} per launch ({n[(k, cn)]} launches)")
PY
