#!/bin/bash
# tools/ab_base.sh [bench args…] — the working-tree library against slr_amd/csrc/variants/libslrhip_base.so (a build of the
# previous commit), alternating, two rounds, on the workloads given by "$@" (default: cornell)
wls=("$@"); [ ${#wls[@]} -eq 0 ] && wls=("cornell")
for round in 1 2; do for v in base new; do
  if [ "$v" = new ]; then unset SLRHIP_LIBRARY; else export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_base.so; fi
  for wl in "${wls[@]}"; do
  timeout -k 10 280 python bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-5s %-24s' % ('$v', '$wl'), 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us'],1) for n in k}, 'bit_exact', d.get('parity',{}).get('bit_exact_fraction'), flush=True)"
  done
done; done
