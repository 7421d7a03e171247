cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
rm -rf gpurun_out/r02/trace_il
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02/trace_il -- python3 scripts/il_trace.py > gpurun_out/r02/trace_il.log 2>&1; echo "trace rc=$?"
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02/trace_il/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "dwt::" in r["Kernel_Name"] or "copy" in r["Kernel_Name"].lower()]
# last forward call: find the last 'k_il_compose' before the first inverse kernel
names = [r["Kernel_Name"].split("(")[0].replace("void dwt::", "") for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith("k_il_compose")]
end = idx[9]  # 10th forward call's compose
start = idx[8] + 1
base = int(rows[start]["Start_Timestamp"])
prev = None
for r, n in zip(rows[start:end + 1], names[start:end + 1]):
    s, e = int(r["Start_Timestamp"]) - base, int(r["End_Timestamp"]) - base
    print("%-50s start %8.1f dur %7.1f gap %s grid %s" % (n[:50], s / 1e3, (e - s) / 1e3, "" if prev is None else "%.1f" % ((s - prev) / 1e3), r["Grid_Size_X"]))
    prev = e
PY
