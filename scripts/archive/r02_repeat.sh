#!/bin/bash
# Same bench, fresh process each time: how much does the rate differ from process to process?
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-split --no-single ${BENCH_EXTRA:-} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run $i', d['value'], 'Gs/s', d['ms_per_step'], 'ms/step', d['step_ms_rank0']['min'], d['step_ms_rank0']['median'])"
done
