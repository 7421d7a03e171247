#!/bin/bash
# Kernel timeline of one batched forward call for a shape: bash scripts/trace_shape.sh W H BATCH
export TMPDIR=/tmp
OUT=gpurun_out/r02/trace_shape; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 scripts/levels_shape.py $1 $2 $3 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "dwt::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call: the final group of kernels
tail = rows[-12:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"+{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:8.1f} us  gap {gap:7.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}  {r['Kernel_Name'].split('(')[0][:70]}")
    prev_end = e
PY
