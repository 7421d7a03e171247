#!/bin/bash
# Round 5: kernel trace + PMC passes (one counter group per pass, never with other trace domains) over EVERY entry of
# the path (scripts/measure_entries.py: 2-D forward / inverse single image and batches, int 5/3, the interleaved layout,
# 3-D out of place and in place) -- the counters of every kernel that is not the headline.
#   gpurun --timeout 1200 -- 'bash scripts/r06/pmc_kernels.sh'        (last on a box: see scripts/archive/r04/final.sh)
set -u
OUT=gpurun_out/r06/pmc_kernels
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp DWT_HIP_TUNE=1 ENTRIES_NO_HOST=1
ARGS="scripts/measure_entries.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1; echo "sq2 rc=$?"
python3 scripts/r06/pmc_kernels_table.py $OUT > gpurun_out/r06/kernels_pmc.md; head -60 gpurun_out/r06/kernels_pmc.md
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r06/entries_kernel_stats.csv
find $OUT -name "*.csv" -size +24M -delete  # (gpurun merges at most 64 MiB back)
