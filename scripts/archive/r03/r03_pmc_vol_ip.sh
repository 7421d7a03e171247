#!/bin/bash
# Round 3: kernel trace + PMC passes (one counter group per pass, never with other trace domains) of the
# in-place 3-D level, 1024^3, forward and inverse, one level.
#   gpurun --timeout 900 -- 'bash scripts/archive/r03/r03_pmc_vol_ip.sh'
set -u
OUT=gpurun_out/r03/pmc_vol_ip
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="scripts/archive/r03/r03_vol_ip_variants.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
python3 scripts/pmc_table.py $OUT "dwt::" > gpurun_out/r03/pmc_vol_ip.txt; head -80 gpurun_out/r03/pmc_vol_ip.txt
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r03/vol_ip_kernel_stats.csv
find $OUT -name "*.csv" -size +2M -delete
