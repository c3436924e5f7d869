#!/bin/bash
# tools/profile_r03.sh TAG [quick] — on the GPU box: kernel-trace stats of the default bench command, then PMC passes
#   * FETCH_SIZE and WRITE_SIZE (separate runs, no trace domains) for EACH BASELINE workload at its bench.py default size and
#     pass count, merged into gpurun_out/TAG/pmc_traffic.json (copy to profiles/pmc_traffic.json);
#   * SQ cycle / instruction counters and the vector-L1 (TCP) / texture-addresser (TA) counters in passes of <= 3 counters
#     (round 2's five-counter TA pass was refused by the profiler: error 38, "exceeds the capabilities of the hardware") on the
#     headline and, at 256 spp, on the 10 M-triangle grid -> gpurun_out/TAG/pmc_*.json (tools/pmc_summary.py).
# Every PMC pass runs bench.py WITHOUT kernel timing and without the counting context, so that only the timed launches are
# profiled (round 2's passes also profiled 32 small launches of the 16-spp counting render, which diluted the per-launch bytes).
tag=$1; quick=$2; repo=$PWD; out=$repo/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $repo/bench.py --cpu-seconds 0 > $out/bench_under_rocprof.json 2> $out/stats.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/kernel_stats.csv \; ; rm -rf $out/stats
echo "stats done" >&2
echo "{}" > $out/pmc_traffic.json
B="--cpu-seconds 0 --no-parity --no-kernel-timing --steps 1 --warmup 0"
traffic() { key=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $repo/bench.py $B "$@" > $out/${key}_$c.json 2> $out/${key}_$c.err
    echo "$key $c done" >&2
  done
  python3 $repo/tools/pmc_traffic.py $key $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_traffic.json > $out/pmc_traffic.tmp && mv $out/pmc_traffic.tmp $out/pmc_traffic.json
  rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE; }
counters() { name=$1; wl=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $out/c_$name -- python3 $repo/bench.py $B $wl > $out/c_$name.json 2> $out/c_$name.err
  python3 $repo/tools/pmc_summary.py $out/c_$name > $out/pmc_$name.json; rm -rf $out/c_$name; echo "$name done" >&2; }
traffic cornell_1280x720_1024spp
if [ "$quick" != quick ]; then
  traffic boxes_spectral_1280x720_1024spp --workload boxes_spectral
  traffic ibl_1280x720_2048spp --workload ibl
  traffic grid10m_1280x720_4096spp --workload grid10m
fi
for wl in "cornell:--spp 256" "grid10m:--workload grid10m --spp 256" "boxes_spectral:--workload boxes_spectral --spp 128"; do
  n=${wl%%:*}; a=${wl#*:}
  counters ${n}_sq_cycles "$a" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
  counters ${n}_sq_insts "$a" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  counters ${n}_tcp_a "$a" TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
  counters ${n}_tcp_b "$a" TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
  counters ${n}_ta_a "$a" TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
  counters ${n}_ta_b "$a" TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
  counters ${n}_ta_c "$a" TA_TOTAL_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
  counters ${n}_tcp_c "$a" TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  counters ${n}_tcp_d "$a" TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
  counters ${n}_tcc "$a" TCC_HIT_sum TCC_MISS_sum
  counters ${n}_grbm "$a" GRBM_GUI_ACTIVE
done
echo "all done" >&2
