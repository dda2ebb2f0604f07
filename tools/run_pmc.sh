#!/bin/bash
# GPU box: PMC counters for the step kernel, one counter group per rocprofv3 pass (never combined with
# --sys-trace etc.).  Usage: tools/run_pmc.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$tag
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-env-leg $BENCH_ARGS > $O/$name.json 2> $O/$name.err; }
BENCH_ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
# fabric traffic of the L2 (TCC -> EA requests by size: exact bytes, no FETCH_SIZE correction needed), L2 hit / miss
run tcc_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run tcc_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
run tcc_dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_REQ_sum TCC_READ_sum
find $O -name "*counter_collection.csv" | head
