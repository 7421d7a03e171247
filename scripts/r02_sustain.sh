#!/bin/bash
# Burst against sustained rate: the same bench at growing step counts (one process each).
for k in 20 60 200 600 20; do
  timeout -k 10 300 python bench.py --steps $k --warmup 5 --no-cpu --no-split --no-single 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('steps $k', d['value'], 'Gs/s', d['ms_per_step'], 'ms/step', d['step_ms_rank0'])"
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk" | head -4
done
