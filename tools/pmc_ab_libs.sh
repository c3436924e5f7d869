#!/bin/bash
# tools/pmc_ab_libs.sh TAG "COUNTERS_PASS1" ["COUNTERS_PASS2" ...] — counter TOTALS per frame of the hot kernels for the working-tree
# library and for slr_amd/csrc/variants/libslrhip_base.so on the same box (one PMC pass per counter set and library, no trace
# domains): gpurun_out/TAG_{new,base}_N.json (tools/pmc_summary.py).  Workload: the headline at 256 spp.
tag=$1; shift; repo=$PWD; out=$repo/gpurun_out; cd /tmp && export TMPDIR=/tmp
B="--cpu-seconds 0 --no-parity --no-kernel-timing --steps 1 --warmup 0 --spp 256"
n=0
for set in "$@"; do n=$((n+1))
  for v in new base; do
    if [ "$v" = new ]; then unset SLRHIP_LIBRARY; else export SLRHIP_LIBRARY=$repo/slr_amd/csrc/variants/libslrhip_base.so; fi
    rocprofv3 --pmc $set --output-format csv -d $out/${tag}_c -- python3 $repo/bench.py $B > /dev/null 2> $out/${tag}_${v}_$n.err
    python3 $repo/tools/pmc_summary.py $out/${tag}_c > $out/${tag}_${v}_$n.json; rm -rf $out/${tag}_c
    echo "$v set $n done" >&2
  done
done
