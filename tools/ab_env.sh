#!/bin/bash
# tools/ab_env.sh VAR val1 val2 … — alternate values of one environment switch on the bench workloads (run on the GPU box)
var=$1; shift
for round in 1 2; do for v in "$@"; do
  export $var=$v
  for wl in "cornell" "ibl --spp 512" "grid10m --spp 512" "boxes_spectral --spp 256"; do
  timeout -k 10 280 python bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-8s %-18s' % ('$v', '$wl'), 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us']) for n in k if k[n]['launches']}, flush=True)"
  done
done; done
