#!/bin/bash
# tools/refresh_profiles.sh TAG — on the GPU box: kernel-trace stats of the default bench command, the two PMC passes for
# HBM traffic (FETCH_SIZE and WRITE_SIZE in separate runs, no trace domains), then the un-profiled headline and the other
# three BASELINE configs.  Everything lands in gpurun_out/refresh_TAG/; copy what is to be judged into profiles/.
set -e
tag=$1; out=$PWD/gpurun_out/refresh_$tag; mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $repo/bench.py --cpu-seconds 0 > $out/bench_under_rocprof.json 2> $out/stats.err
echo "stats done" >&2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $repo/bench.py --cpu-seconds 0 --no-parity > $out/pmc_fetch.json 2> $out/pmc_fetch.err
echo "fetch done" >&2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $repo/bench.py --cpu-seconds 0 --no-parity > $out/pmc_write.json 2> $out/pmc_write.err
echo "write done" >&2
cd $repo
python3 tools/pmc_traffic.py cornell_1280x720_1024spp $out/pmc_fetch $out/pmc_write > $out/pmc_traffic.json
cp $out/pmc_traffic.json profiles/pmc_traffic.json
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/stats/*/*kernel_trace.csv $out/pmc_fetch $out/pmc_write 2>/dev/null || true
python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "headline done" >&2
python3 bench.py --workload boxes_spectral --steps 1 --warmup 1 > $out/bench_boxes_spectral.json 2> $out/b2.err
python3 bench.py --workload ibl --steps 1 --warmup 1 > $out/bench_ibl.json 2> $out/b3.err
echo "configs 2,3 done" >&2
python3 bench.py --workload grid10m --steps 1 --warmup 0 > $out/bench_grid10m.json 2> $out/b4.err
echo "all done" >&2
