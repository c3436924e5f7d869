#!/bin/bash
# sweep of the wave-specialised traversal's refill threshold on three workloads (run on the GPU box)
for round in 1 2; do for rf in 8 12 16 20 24 32; do
  export SLRHIP_WS_REFILL=$rf
  for wl in "cornell" "ibl --spp 256" "grid10m --spp 64"; do
  timeout -k 10 200 python bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 --no-parity 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('refill %-3s %-18s' % (os.environ['SLRHIP_WS_REFILL'], '$wl'), 'Msamples/s %8.1f' % d['value'], 'trace %.1f us' % k['trace']['avg_us'], flush=True)"
  done
done; done
