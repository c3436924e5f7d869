#!/bin/bash
# sweep of the wave-specialised traversal knobs on the headline workload (run on the GPU box)
export SLRHIP_TRACE=ws
for nc in 3 7; do for rf in 16; do
  export SLRHIP_WS_NC=$nc SLRHIP_WS_REFILL=$rf
  timeout -k 10 120 python bench.py --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('nc', os.environ['SLRHIP_WS_NC'], 'refill', os.environ['SLRHIP_WS_REFILL'], 'Msamples/s', d['value'], {n:k[n]['avg_us'] for n in k}, flush=True)"
done; done
