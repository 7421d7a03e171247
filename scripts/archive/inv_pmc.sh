#!/bin/bash
# HBM traffic and duration of the inverse level-0 kernel (batch of 8 images of 8192^2).
export TMPDIR=/tmp
OUT=gpurun_out/r02/inv_pmc; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, libdwt_amd as dwt
nb, n, J = 8, 8192, 5
dwt.dwt_util_init(); dwt.use_torch_stream()
x = torch.rand((nb, n, n), device="cuda"); y = torch.empty_like(x)
for _ in range(6):
    dwt.transform2d_batch("cdf97_s", 1, x, y, n * n * 4, nb, n * 4, n, n, J)
    dwt.transform2d_batch("cdf97_s", 0, x, y, n * n * 4, nb, n * 4, n, n, J)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $OUT/run.py > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $OUT/run.py > $OUT/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $OUT/run.py > $OUT/s.log 2>&1
python scripts/pmc_table.py $OUT k_ | head -60
