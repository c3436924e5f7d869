#!/bin/bash
# tools/pmc_traffic_one.sh KEY [bench args] — FETCH_SIZE / WRITE_SIZE passes (separate runs, no trace domains) of one bench command,
# merged into profiles/pmc_traffic.json under KEY (the key bench.py looks up: <workload>[_instanced]_<W>x<H>_<spp>spp)
key=$1; shift; repo=$PWD; out=$repo/gpurun_out/pmc_one; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="--cpu-seconds 0 --no-parity --no-kernel-timing --steps 1 --warmup 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $repo/bench.py $B "$@" > $out/${key}_$c.json 2> $out/${key}_$c.err
  echo "$key $c done" >&2
done
python3 $repo/tools/pmc_traffic.py $key $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $repo/profiles/pmc_traffic.json > $out/pmc_traffic.json && cp $out/pmc_traffic.json $repo/gpurun_out/pmc_traffic_merged.json
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
