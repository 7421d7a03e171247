#!/usr/bin/env python3
"""Kernel timeline of single calls from a rocprofv3 kernel trace: calls are separated by gaps > 150 us.
   python scripts/archive/r04/calls_timeline.py <trace dir> <call index> [<call index> ...]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "dwt::" in r["Kernel_Name"]]
calls = [[rows[0]]]
for p, r in zip(rows, rows[1:]):
    if int(r["Start_Timestamp"]) - int(p["End_Timestamp"]) > int(__import__("os").environ.get("GAP_NS", 40000)):
        calls.append([])
    calls[-1].append(r)
print(len(calls), "calls")
for k in [int(x) for x in sys.argv[2:]]:
    c = calls[k]
    t0 = int(c[0]["Start_Timestamp"])
    for r in c:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("void dwt::", "")[:64]
        print(f"{s/1e3:8.1f} .. {e/1e3:8.1f} ({(e-s)/1e3:6.1f}) q{r.get('Queue_Id', '?')} {name} {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
    print("---- total", (int(c[-1]["End_Timestamp"]) - t0) / 1e3, "us")
