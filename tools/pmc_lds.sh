R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_lds
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/a -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-env-leg > $O/a.json 2> $O/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH SQ_ITEMS --output-format csv -d $O/b -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-env-leg > $O/b.json 2> $O/b.err
cd $R; python3 tools/parse_pmc.py lds_x 2>/dev/null; python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_lds/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "wf_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(k, sum(v)/len(v))
PY
tail -3 $O/a.err $O/b.err
