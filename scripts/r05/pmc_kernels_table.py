#!/usr/bin/env python3
"""One table of every kernel of scripts/measure_entries.py from the rocprofv3 passes of scripts/r05/pmc_kernels.sh:
    python scripts/r05/pmc_kernels_table.py gpurun_out/r05/pmc_kernels > profiles/r05_kernels_pmc.md
Per (kernel, grid): launches, average duration, HBM-side traffic per launch (FETCH_SIZE doubled -- the gfx950 correction
of MI355X_MICROARCH.md for wide streaming reads -- plus WRITE_SIZE; both counters are in KiB), L2 hit rate, the share of
VALU issue cycles in the busy cycles, LDS bank-conflict cycles per LDS instruction cycle; and, where the launch's shape is
known (ALG below), traffic / algorithmic bytes."""
import collections, csv, glob, re, sys

src = sys.argv[1]


def short(n):
    n = n.split("(")[0].replace("void ", "").replace("dwt::", "").replace("dwtb::", "")
    return re.sub(r"\s+", "", n)


dur = collections.defaultdict(list)
for f in glob.glob(f"{src}/trace/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r.get("Grid_Size_Z", 1) or 1))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))

# algorithmic bytes per launch of the shapes measure_entries.py runs: (kernel substring, grid in threads) -> bytes.
# 2-D level j of B images of n x n: 8 B x B x (n >> j)^2; 3-D level j of n^3: 8 B x (n >> j)^3; copies: 2 x payload.
ALG = {}


def alg2d(B, n, j):
    return 8 * B * (n >> j) ** 2


def mean(v):
    v = v[2:] if len(v) > 4 else v  # (the first launches of a shape: warm-up, tile tuner)
    return sum(v) / len(v) if v else 0.0


rows = []
for key in set(dur) | set(acc):
    if not any(s in key[0] for s in ("k_", "sweep", "vol", "lattice", "shell", "copy")):
        continue
    d, c = dur.get(key, []), acc.get(key, {})
    fetch = mean(c.get("FETCH_SIZE", [])) * 1024 * 2
    write = mean(c.get("WRITE_SIZE", [])) * 1024
    hit, miss = mean(c.get("TCC_HIT_sum", [])), mean(c.get("TCC_MISS_sum", []))
    busy, valu = mean(c.get("SQ_BUSY_CYCLES", [])), mean(c.get("SQ_ACTIVE_INST_VALU", []))
    wavec = mean(c.get("SQ_WAVE_CYCLES", []))
    ldsc, ldsa = mean(c.get("SQ_LDS_BANK_CONFLICT", [])), mean(c.get("SQ_ACTIVE_INST_LDS", []))
    rows.append((sum(d), key, len(d), mean(d) / 1e3, fetch, write, hit / (hit + miss) if hit + miss else None,
                 valu / wavec if wavec else None, ldsc / ldsa if ldsa else None, mean(c.get("SQ_INSTS_VALU", []))))
rows.sort(reverse=True)
print("| kernel | grid (threads) | launches | avg us | fetch MB (x2) | write MB | traffic GB/s | L2 hit | VALU active / wave cycles | LDS conflict / LDS active | VALU insts (M) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for tot, key, n, us, fetch, write, hit, valu, lds, nvalu in rows:
    if us <= 0 or n == 0:
        continue
    f = lambda x, p=2: "" if x is None else f"{x:.{p}f}"
    gbs = (fetch + write) / (us * 1e-6) / 1e9 if fetch + write else 0
    print(f"| `{key[0][:90]}` | {key[1]} | {n} | {us:.1f} | {fetch / 1e6:.1f} | {write / 1e6:.1f} | {gbs:.0f} | {f(hit)} | {f(valu)} | {f(lds, 3)} | {nvalu / 1e6:.2f} |")
