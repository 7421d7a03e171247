#!/bin/bash
# 3-D out-of-place: levels >= 1 written straight into their lattice vs dense + scatter.
set -u
OUT=gpurun_out/r02/vol_direct; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_hip_volume.py -x -q -m gpu > $OUT/test.log 2>&1; rc=$?
tail -3 $OUT/test.log
[ $rc -eq 0 ] || exit $rc
VARIANTS="vol_nt=-1;vol_nt=3;vol_direct=1;vol_whole=0;vol_nt=-1;vol_nt=3;vol_direct=1;vol_whole=0;vol_direct=0,vol_whole=0" timeout -k 10 200 python scripts/vol_op_bench.py 1024 3 2>&1 | tee $OUT/bench.log
VARIANTS="vol_nt=-1;vol_nt=3;vol_nt=-1;vol_nt=3" timeout -k 10 200 python scripts/vol_op_bench.py 512 3 2>&1 | tee -a $OUT/bench.log
timeout -k 10 200 python scripts/vol_ip_bench.py 2>&1 | tee -a $OUT/bench.log
