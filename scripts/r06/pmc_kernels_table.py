#!/usr/bin/env python3
"""One table of every kernel of scripts/measure_entries.py from the rocprofv3 passes of scripts/r06/pmc_kernels.sh:
    python scripts/r06/pmc_kernels_table.py gpurun_out/r06/pmc_kernels > profiles/r06_kernels_pmc.md
The workload is deterministic, so the passes are joined PER DISPATCH (Dispatch_Id; a dispatch whose kernel name differs
between passes -- a tile height the tuner chose differently -- is dropped).  Dispatches are then grouped by (kernel, grid,
megabytes written): the written bytes pin the shape, so launches of one kernel on different levels / sizes stay apart.
Columns: launches, average duration (kernel-trace pass), HBM-side traffic per launch = FETCH_SIZE x 2 (the gfx950
correction of MI355X_MICROARCH.md for wide streaming reads; uncalibrated for the 4-byte halo pieces) + WRITE_SIZE (both
counters are in KiB), the rate of that traffic, L2 hit rate, VALU-active share of the wave cycles, LDS bank-conflict
cycles per LDS-active cycle, VALU instructions; `alg` = algorithmic bytes where the shape is recognised (ALG below),
`x alg` = traffic / algorithmic."""
import collections, csv, glob, re, sys

src = sys.argv[1]


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("dwt::", "").replace("dwtb::", "")
    return re.sub(r"\s+", "", n)


disp = {}  # dispatch id -> {"name", "grid", "us", counters...}
for f in glob.glob(f"{src}/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        disp[int(r["Dispatch_Id"])] = {"name": short(r["Kernel_Name"]), "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                       "grid": int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), "wg": int(r["Workgroup_Size_X"]), "ok": True}
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d = disp.get(int(r["Dispatch_Id"]))
        if d is None:
            continue
        if d["name"] != short(r["Kernel_Name"]) or d["grid"] != int(r["Grid_Size"]):
            d["ok"] = False
            continue
        d[r["Counter_Name"]] = float(r["Counter_Value"])

groups = collections.defaultdict(list)
for i, d in sorted(disp.items()):
    if not d["ok"] or "WRITE_SIZE" not in d or not any(s in d["name"] for s in ("k_", "sweep", "vol", "lattice", "shell", "copy_rects")):
        continue
    groups[(d["name"], d["grid"], round(d["WRITE_SIZE"] / 1024))].append(d)

MB = 1e6
# algorithmic bytes of the launches the workload is known to make: kernel prefix -> {written MB (rounded MiB key) -> (bytes, what)}
def alg_of(name, wmib, grid):
    n = 8192
    t = []
    if name.startswith(("k_fwd_sweep<Cdf97S", "k_inv_sweep<Cdf97S")) and "true>" not in name:
        for B in (1, 8):
            for j in range(5):
                t.append((4 * B * (n >> j) ** 2, 8 * B * (n >> j) ** 2, f"2-D level {j}, {B} image(s) of 8192^2"))
        for j in range(5):
            t.append((4 * 32 * (4096 >> j) ** 2, 8 * 32 * (4096 >> j) ** 2, f"2-D level {j}, 32 images of 4096^2"))
    if name.startswith(("k_fwd_sweep<Cdf53I", "k_inv_sweep<Cdf53I")):
        for B in (1, 16):
            for j in range(3):
                t.append((4 * B * (4096 >> j) ** 2, 8 * B * (4096 >> j) ** 2, f"int 5/3 level {j}, {B} image(s) of 4096^2"))
    if name.startswith(("k_vol_level_ip", "k_vol_fwd_fused", "k_vol_z")):
        for nn in (1024, 512):
            for j in range(3):
                t.append((4 * (nn >> j) ** 3, 8 * (nn >> j) ** 3, f"3-D level {j} of {nn}^3"))
    for w, a, what in t:
        if abs(w / 2 ** 20 - wmib) <= max(1.0, 0.06 * wmib):
            return a, what
    return None, ""


rows = []
for (name, grid, wmib), ds in groups.items():
    ds = ds[2:] if len(ds) > 4 else ds  # (first launches of a shape: warm-up, tile tuner)
    m = lambda k: sum(d.get(k, 0.0) for d in ds) / len(ds)
    fetch, write = m("FETCH_SIZE") * 1024 * 2, m("WRITE_SIZE") * 1024
    hit, miss = m("TCC_HIT_sum"), m("TCC_MISS_sum")
    rows.append((m("us") * len(ds), name, grid, len(ds), m("us"), fetch, write, hit / (hit + miss) if hit + miss else None,
                 m("SQ_ACTIVE_INST_VALU") / m("SQ_WAVE_CYCLES") if m("SQ_WAVE_CYCLES") else None,
                 m("SQ_LDS_BANK_CONFLICT") / m("SQ_ACTIVE_INST_LDS") if m("SQ_ACTIVE_INST_LDS") else None, m("SQ_INSTS_VALU"), wmib))
rows.sort(reverse=True)
print("| kernel | grid (threads) | shape | launches | avg us | fetch MB (x2) | write MB | traffic GB/s | alg MB | x alg | alg GB/s | L2 hit | VALU active | LDS conflict | VALU insts (M) |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
f = lambda x, p=2: "" if x is None else f"{x:.{p}f}"
for tot, name, grid, n, us, fetch, write, hit, valu, lds, nvalu, wmib in rows:
    if us < 8:
        continue
    alg, what = alg_of(name, wmib, grid)
    tr = fetch + write
    print(f"| `{name[:80]}` | {grid} | {what} | {n} | {us:.1f} | {fetch / MB:.1f} | {write / MB:.1f} | {tr / us / 1e3:.0f} | {f(alg / MB if alg else None, 1)} | "
          f"{f(tr / alg if alg else None)} | {f(alg / us / 1e3 if alg else None, 0)} | {f(hit)} | {f(valu)} | {f(lds, 3)} | {nvalu / 1e6:.2f} |")
