#!/bin/bash
# The library (the reference's line-end forms in every 2-D float kernel) against `make plain` (reflected ends), alternated on
# ONE box: whole 2-D calls (ab_calls.py) and the bench line.  Needs libdwt_amd/libdwt_hip_plain.so (make -C libdwt_amd/csrc plain).
#   gpurun --timeout 1100 -- 'bash scripts/r06/end_forms_ab.sh > gpurun_out/end_forms_ab.txt 2>&1'
set -u
L=$PWD/libdwt_amd
for r in 1 2 3; do
  for v in hip_plain hip; do
    DWT_HIP_LIB=$L/libdwt_$v.so python scripts/r06/ab_calls.py 2>&1 | grep -v amdgpu.ids
  done
done
for r in 1 2; do
  for v in hip_plain hip; do
    echo "bench $v: $(DWT_HIP_LIB=$L/libdwt_$v.so python bench.py --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"], "batch_inverse", d.get("extra_legs", {}).get("batch_inverse", {}).get("value"))')"
  done
done
