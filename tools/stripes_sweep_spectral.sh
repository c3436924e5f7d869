export SLRHIP_TAIL_SLOTS=262144
for round in 1 2; do for k in 4 8 16 32; do
  timeout -k 10 280 python bench.py --workload boxes_spectral --spp 512 --stripes $k --cpu-seconds 0 --no-parity --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('K=%-3s spectral' % '$k', 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us']) for n in k if k[n]['launches']}, flush=True)"
done; done
