#!/bin/bash
for im in 16 16 16 16 16 16 64 64 64 64 64 64; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --images $im --no-cpu --no-split --no-single 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('images $im', d['value'], 'Gs/s', d['ms_per_step'], 'ms/step', d['step_ms_rank0']['min'], d['step_ms_rank0']['median'])"
done
