#!/bin/bash
# tools/ab_variants.sh [variant names…] — alternate default and variant builds on the headline and the 10 M-triangle workloads
# (run on the GPU box; two rounds so that box/clock drift shows up as spread)
for round in 1 2; do for v in default "$@"; do
  if [ "$v" = default ]; then unset SLRHIP_LIBRARY; else export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_$v.so; fi
  for wl in "cornell" "grid10m --spp 64"; do
  timeout -k 10 280 python bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-10s %-18s' % ('$v', '$wl'), 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us']) for n in k}, flush=True)"
  done
done; done
