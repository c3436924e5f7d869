#!/bin/bash
# tools/pmc_trace_probe.sh TAG [env assignments...] — PMC passes on the headline bench (256 spp, 1 step) to see what bounds
# k_trace_ws: wave-cycle breakdown, instruction mix, LDS, and the texture-addresser / vector-L1 side.  Summaries -> gpurun_out/TAG_*.json
tag=$1; shift
for kv in "$@"; do export "$kv"; done
repo=$PWD; out=$repo/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 $repo/bench.py --cpu-seconds 0 --no-parity --steps 1 --warmup 0 --spp 256 > $out/$name.json 2> $out/$name.err
  python3 $repo/tools/pmc_summary.py $out/$name > $repo/gpurun_out/${tag}_$name.json; rm -rf $out/$name; echo "$name done" >&2; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# (the TA_* / TCP_* passes that were here never finished on this pool: the run sat silent until the 7-minute watchdog killed it; the SQ
#  passes above are what DESIGN 8.2 quotes)
