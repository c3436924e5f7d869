export SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_tailall.so SLRHIP_TAIL_SLOTS=100000000
for k in 1 4 32; do
timeout -k 10 250 python bench.py --stripes $k --spp 256 --cpu-seconds 0 --no-parity --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('tail kernel as the renderer, stripes $k:', 'Msamples/s %8.1f' % d['value'], {n:(k[n]['launches'], round(k[n]['ms_total'],1)) for n in k}, flush=True)"
done
