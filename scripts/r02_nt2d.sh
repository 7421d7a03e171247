#!/bin/bash
# 2-D forward sweep: cache policy of the loads, re-measured after the 3-D finding.
mkdir -p gpurun_out/r02
for o in "" "--opt nt=5" "--opt nt=7" "--opt nt=5" "--opt nt=4" "--opt nt=1"; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-split $o 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$o', d['value'], d['ms_per_step'], d.get('single_image',{}).get('s2_us_median'), d['roofline']['frac'])"
done
