#!/bin/bash
# tools/ab_trees.sh "WORKLOAD ARGS" [rounds] — same-box A/B of the working tree against the round-2 tree kept (built, git-ignored) under
# slr_amd/csrc/variants/r02_tree: each runs its OWN bench.py, alternating, so that box / clock drift shows up as spread.
# Extra environment for the working tree's runs: NEW_ENV="SLRHIP_AUTO_STRIPES=32".
wl=$1; rounds=${2:-2}
root=$PWD
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernels',{})
print('%-14s %-28s' % ('$1', '$wl'), 'Msamples/s %8.1f' % d['value'], 'ms/step %8.2f' % d['ms_per_step'], {n:round(k[n]['avg_us'],1) for n in k}, 'iters', d.get('counters',{}).get('iterations'), flush=True)"; }
for r in $(seq $rounds); do
  (cd $root/slr_amd/csrc/variants/r02_tree && timeout -k 10 280 python3 bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | line r02)
  (cd $root && env $NEW_ENV timeout -k 10 280 python3 bench.py --workload $wl --cpu-seconds 0 --steps 1 --warmup 1 2>/dev/null | line "new $NEW_ENV")
done
