#!/bin/bash
# A/B of the two traversal schedules on every bench workload (run on the GPU box)
for wl in "cornell" "ibl --spp 512" "grid10m --spp 128" "boxes_spectral --spp 256"; do for mode in batch ws; do
  export SLRHIP_TRACE=$mode
  timeout -k 10 280 python bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$wl', os.environ['SLRHIP_TRACE'], 'Msamples/s', d['value'], {n:k[n]['avg_us'] for n in k}, flush=True)"
done; done
