#!/bin/bash
# Per process: level-0 kernel duration and address-translation counters.
export TMPDIR=/tmp
OUT=gpurun_out/r02/tlb; rm -rf $OUT; mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --output-format csv -d $OUT/p$i -- python3 bench.py --steps 5 --warmup 2 --images 16 --no-cpu --no-split --no-single > $OUT/p$i.log 2>&1
  python3 - $OUT/p$i <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = []; acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fwd_sweep" in r["Kernel_Name"] and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) >= 2048 * 256 * 16:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_fwd_sweep" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 2048 * 256 * 16:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(d, "level-0 launches", len(dur), "avg us", round(sum(dur) / max(len(dur), 1) / 1e3, 1), {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
done
