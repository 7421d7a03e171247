# usage: bash scripts/r02_pmc_vol.sh <tag> <variant string>
cd $GRAFT_REPO_ROOT
TAG=$1; export VARIANTS="$2"
OUT=gpurun_out/r02/pmcvol_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="scripts/vol_op_bench.py 1024 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > $OUT/pmc_sq2.log 2>&1; echo "sq2 rc=$?"
python scripts/pmc_table.py $OUT k_vol > gpurun_out/r02/pmcvol_$TAG.txt; cat gpurun_out/r02/pmcvol_$TAG.txt | head -60
find $OUT -name "*.csv" -size +2M -delete
