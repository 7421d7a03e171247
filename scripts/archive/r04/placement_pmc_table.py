#!/usr/bin/env python3
"""Per level-0 dispatch of each pass of placement_pmc.sh: duration (kernel trace) and counters."""
import csv, glob, sys, collections
src = sys.argv[1]
for tag in ("utcl", "stall", "level", "lat", "req"):
    dur = {}
    for f in glob.glob(f"{src}/{tag}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "k_fwd_sweep" in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))
    ctr = collections.defaultdict(dict)
    for f in glob.glob(f"{src}/{tag}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_fwd_sweep" in r["Kernel_Name"]:
                ctr[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    if not dur:
        print(f"## {tag}: no data"); continue
    big = max(g for _, g in dur.values())
    names = sorted({c for d in ctr.values() for c in d})
    print(f"## pass {tag}: level-0 dispatches (4 per placement, in order)")
    print("dispatch  ms      " + "  ".join(f"{c:>40s}" for c in names))
    for d in sorted(dur):
        if dur[d][1] != big: continue
        print(f"{d:8d} {dur[d][0]/1e6:6.3f}  " + "  ".join(f"{ctr[d].get(c, float('nan')):40.0f}" for c in names))
