#!/bin/bash
# Per-kernel durations of the in-place 3-D calls (1024^3, forward + inverse, 1 and 3 levels).
#   gpurun -- 'bash scripts/archive/r03/r03_vol_ip_trace.sh'
set -u
OUT=gpurun_out/r03/vol_ip_trace; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 scripts/archive/vol_ip_bench.py ${1:-1024} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cat $OUT/run.log | grep "in place"
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/r03/vol_ip_kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
