#!/bin/bash
# paths in flight per pixel (sample stripes) on the headline, the environment-light and the spectral workloads
for wl in "cornell" "ibl" "boxes_spectral --spp 256"; do for st in 4 8 12 16 24 32; do
  timeout -k 10 200 python bench.py --workload $wl --stripes $st --cpu-seconds 0 --steps 1 --warmup 1 --no-parity 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-24s stripes %-3s' % ('$wl', '$st'), 'Msamples/s %8.1f' % d['value'], {n:(round(k[n]['avg_us']), k[n]['launches']) for n in k}, flush=True)"
done; done
