#!/bin/bash
for o in "--warmup 5" "--warmup 5" "--warmup 30" "--warmup 5" ; do
  timeout -k 10 200 python bench.py --steps 20 $o --no-cpu --no-split --no-single 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$o', d['value'], d['ms_per_step'], d['step_ms_rank0'] if 'step_ms_rank0' in d else '')"
done
