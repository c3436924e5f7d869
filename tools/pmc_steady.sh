#!/bin/bash
# tools/pmc_steady.sh TAG [bench args] — per-dispatch FETCH_SIZE / WRITE_SIZE of one render (separate PMC passes, no trace
# domains), written as a table in launch order to gpurun_out/TAG_per_dispatch.txt
tag=$1; shift; repo=$PWD; out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="--cpu-seconds 0 --no-parity --no-kernel-timing --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $repo/bench.py $B "$@" > $out/$c.json 2> $out/$c.err
  echo "$c done" >&2
done
python3 $repo/tools/pmc_per_dispatch.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $repo/gpurun_out/${tag}_per_dispatch.txt
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
