#!/bin/bash
# tools/ab_named.sh "WORKLOAD ARGS" VARIANT… — the working-tree library ("default") and the named variants of
# slr_amd/csrc/variants/ alternating on one workload, two rounds
wl=$1; shift
for round in 1 2; do for v in default "$@"; do
  if [ "$v" = default ]; then unset SLRHIP_LIBRARY; else export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_$v.so; fi
  timeout -k 10 280 python bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-8s %-24s' % ('$v', '$wl'), 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us'],1) for n in k}, 'exact', d.get('parity',{}).get('bit_exact_fraction'), flush=True)"
done; done
