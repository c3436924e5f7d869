for round in 1 2; do for v in base new; do
  if [ "$v" = new ]; then unset SLRHIP_LIBRARY; else export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_$v.so; fi
  timeout -k 10 280 python bench.py --workload grid10m --instanced --spp 512 --cpu-seconds 0 --no-parity --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-9s' % '$v', 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us'],1) for n in k}, flush=True)"
done; done
