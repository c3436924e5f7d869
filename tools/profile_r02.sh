#!/bin/bash
# tools/profile_r02.sh TAG — on the GPU box: kernel-trace stats of the default bench command, then the two PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, no trace domains) for EACH of the four BASELINE workloads, merged into
# gpurun_out/TAG/pmc_traffic.json (copy to profiles/pmc_traffic.json).  Progress lines go to stderr every pass.
tag=$1; repo=$PWD; out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $repo/bench.py --cpu-seconds 0 > $out/bench_under_rocprof.json 2> $out/stats.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; ; rm -rf $out/stats
echo "stats done" >&2
: > $out/pmc_traffic.json; echo "{}" > $out/pmc_traffic.json
pmc() { key=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $repo/bench.py --cpu-seconds 0 --no-parity --steps 1 --warmup 0 "$@" > $out/${key}_$c.json 2> $out/${key}_$c.err
    echo "$key $c done" >&2
  done
  python3 $repo/tools/pmc_traffic.py $key $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_traffic.json > $out/pmc_traffic.tmp && mv $out/pmc_traffic.tmp $out/pmc_traffic.json
  rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE; }
pmc cornell_1280x720_1024spp
pmc boxes_spectral_1280x720_256spp --workload boxes_spectral --spp 256
pmc ibl_1280x720_512spp --workload ibl --spp 512
pmc grid10m_1280x720_256spp --workload grid10m --spp 256
echo "all done" >&2
