#!/bin/bash
# Per-kernel durations of the 3-level 1024^3 out-of-place call, per store variant.
set -u
OUT=gpurun_out/r02/vol_trace; mkdir -p $OUT
export TMPDIR=/tmp
for v in ${DIRECTS:-2 1 0}; do
  VARIANTS="vol_direct=$v" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d$v -- python3 scripts/vol_op_bench.py 1024 3 > $OUT/d$v.log 2>&1 || exit 1
  f=$(find $OUT/d$v -name "*kernel_stats.csv" | head -1)
  echo "== vol_direct=$v"; cat $OUT/d$v.log | grep level; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
