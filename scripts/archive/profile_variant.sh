#!/bin/bash
# PMC profile of the level-0 launch for one option set: bash scripts/profile_variant.sh tag "fuse2=1"
TAG=$1; OPTS=$2
OUT=gpurun_out/pv_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu"
for o in $(echo $OPTS | tr ',' ' '); do ARGS="$ARGS --opt $o"; done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $ARGS > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $ARGS > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/w -- python3 $ARGS > $OUT/w.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ["sq","sq2","f","w"]:
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % d)
    if not fs: print(d, "none"); continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "sweep" in r["Kernel_Name"]]
    g = max(int(r["Grid_Size"]) for r in rows)
    acc = collections.defaultdict(list)
    name = ""
    for r in rows:
        if int(r["Grid_Size"]) == g:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); name = r["Kernel_Name"][:60]
    print(d, name, g, {k: round(sum(v)/len(v)) for k, v in acc.items()})
PY
