import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last forward call: find sequences; print the kernels of one call in the middle
names = [r["Kernel_Name"][:70] for r in rows]
# a call starts with k_fwd_sweep on largest grid
calls = []
cur = []
for r in rows:
    if "k_fwd_sweep" in r["Kernel_Name"] and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) >= 8192*8192//64*0 and cur and "k_il_compose" in cur[-1]["Kernel_Name"]:
        calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
c = calls[5] if len(calls) > 6 else calls[-1]
t0 = int(c[0]["Start_Timestamp"])
for r in c:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:8.1f} .. {e/1e3:8.1f} us  ({(e-s)/1e3:6.1f})  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:60]} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
