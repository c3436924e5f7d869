#!/bin/bash
# tools/bench_counts.sh [bench.py args] — one line: throughput, kernel averages, nodes / triangles per ray, tree size and build time
python3 bench.py --cpu-seconds 0 --no-parity "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']; r=d['roofline']['per_ray']; c=d['counters']
print('%.1f Msamples/s  ' % d['value'] + '  '.join('%s %.1f us' % (n, k[n]['avg_us']) for n in k) +
      '  | closest %.2f nodes %.2f tris, shadow %.2f nodes %.2f tris | %d nodes depth %d build %.2f s' % (
      r['nodes_closest'], r['tris_closest'], r['nodes_shadow'], r['tris_shadow'], c['bvh_nodes'], c['bvh_depth'], c['build_seconds']))"
