#!/bin/bash
# Fast / slow placement under counters: scripts/archive/probes/r04_placement_run.py under one rocprofv3 pass per
# counter group, each with --kernel-trace so that every level-0 dispatch has its duration AND its
# counters (placements differ from dispatch to dispatch inside one process).
#   gpurun --timeout 900 -- 'bash scripts/archive/r04/placement_pmc.sh'
set -u
OUT=gpurun_out/r04_placement
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$tag -- python3 scripts/archive/probes/r04_placement_run.py > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"; }
run utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
run stall TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
run level TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUBBLE_sum
run lat TCC_READ_REQ_LATENCY_sum TCC_WRITE_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_WRITE_REQ_sum
run req TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
python3 scripts/archive/r04/placement_pmc_table.py $OUT > $OUT/table.txt 2>&1
cat $OUT/table.txt
