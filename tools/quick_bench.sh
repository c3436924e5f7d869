#!/bin/bash
# tools/quick_bench.sh [bench.py args] — headline bench without the CPU legs, one line of kernel averages
python3 bench.py --cpu-seconds 0 --no-parity "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%.1f Msamples/s  %.1f ms/step  ' % (d['value'], d['ms_per_step']) + '  '.join('%s %.1f us' % (n, k[n]['avg_us']) for n in k if k[n]['avg_us'] is not None))"
