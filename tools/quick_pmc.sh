#!/bin/bash
# GPU box: the counters the round's kernel work is steered by, three rocprofv3 passes around a short headline bench
# (never combined with --sys-trace etc.).  Usage: tools/quick_pmc.sh <tag> [bench args] -> gpurun_out/<tag>_pmc.json
# (WF_CALIBRATE=0: the handle's own timing of the other kernel families would put their launches under the same counters)
tag=$1; shift
export WF_CALIBRATE=0
KERNEL=${PMC_KERNEL:-wf_step_ll_kernel}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH_ARGS="$*"
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-env-leg $BENCH_ARGS > $O/$name.json 2> $O/$name.err; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run tcc_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run tcc_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
cd $R
python3 tools/parse_pmc.py $tag "$KERNEL" > gpurun_out/${tag}_pmc.json
cat gpurun_out/${tag}_pmc.json
rm -rf $O
