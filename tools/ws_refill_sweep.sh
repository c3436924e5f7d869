export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_knobs.so
for round in 1 2; do for r in 8 14 20 28 40; do
  export SLRHIP_WS_REFILL=$r
  for wl in "cornell --spp 512" "grid10m --spp 256"; do
  timeout -k 10 280 python bench.py --workload $wl --cpu-seconds 0 --no-parity --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('refill=%-3s %-18s' % ('$r', '$wl'), 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us']) for n in k if k[n]['launches']}, flush=True)"
  done
done; done
