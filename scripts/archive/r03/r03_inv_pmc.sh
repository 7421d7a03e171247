#!/bin/bash
# HBM traffic and duration of the inverse level-0 kernel (batch of 8 images of 8192^2, ONE level) per option set:
#   bash scripts/archive/r03/r03_inv_pmc.sh "tile_pairs=0" "tile_pairs=64" ...
export TMPDIR=/tmp
for V in "$@"; do
  TAG=$(echo "$V" | tr ',=' '__')
  OUT=gpurun_out/r03/inv_pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
  cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import torch, libdwt_amd as dwt
nb, n, J = 8, 8192, 1
dwt.dwt_util_init(); dwt.use_torch_stream()
for kv in "$V".split(","):
    k, v = kv.split("="); dwt.set_option(k, int(v))
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
for _ in range(8):
    dwt.transform2d_batch("cdf97_s", 1, x, y, n * n * 4, nb, n * 4, n, n, J)
torch.cuda.synchronize()
PY
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $OUT/run.py > $OUT/f.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $OUT/run.py > $OUT/w.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $OUT/run.py > $OUT/s.log 2>&1
  echo "=== $V"; python3 scripts/pmc_table.py $OUT k_inv | head -24
  find $OUT -name "*.csv" -size +2M -delete
done
