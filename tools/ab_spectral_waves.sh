for round in 1 2; do for v in default w3 w1; do
  if [ "$v" = default ]; then unset SLRHIP_LIBRARY; else export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_$v.so; fi
  timeout -k 10 280 python bench.py --workload boxes_spectral --spp 256 --cpu-seconds 0 --no-parity --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-9s' % '$v', 'Msamples/s %8.1f' % d['value'], {n:round(k[n]['avg_us'],1) for n in k}, flush=True)"
done; done
