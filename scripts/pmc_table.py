#!/usr/bin/env python3
"""Per-kernel averages of every counter in the rocprofv3 --pmc passes under <dir>/pmc_*/ (and the
kernel durations from <dir>/trace), largest grid of each kernel name only:
    python scripts/pmc_table.py gpurun_out/r02/pmc_fuse2 [name-substring]"""
import csv, glob, sys, collections
src = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "dwt::"
def short(n): return n.split("(")[0].replace("void ", "").replace("dwt::", "")
dur = collections.defaultdict(list)
for f in glob.glob(f"{src}/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(set(dur) | set(acc), key=lambda k: -k[1]):
    d = dur.get(key, [])
    d = d[2:] if len(d) > 4 else d
    print(f"\n## {key[0]}  grid {key[1]}  launches {len(d)}  avg {sum(d)/max(len(d),1)/1e3:.1f} us")
    for c, v in sorted(acc.get(key, {}).items()):
        v = v[2:] if len(v) > 4 else v
        print(f"   {c:28s} {sum(v)/len(v):16.1f}")
