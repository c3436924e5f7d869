#!/bin/bash
for st in 2 4 6 8 12 16; do
  timeout -k 10 120 python bench.py --stripes $st --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('stripes $st', 'Msamples/s %8.1f' % d['value'], {n:(round(k[n]['avg_us']), k[n]['launches']) for n in k}, flush=True)"
done
