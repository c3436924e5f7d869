#!/bin/bash
# tools/bench_all.sh TAG — the un-profiled bench lines of the four BASELINE workloads (+ the instanced grid) on one box:
# gpurun_out/TAG_bench_{n1,boxes_spectral,ibl,grid10m,grid10m_instanced}.json
tag=$1
python3 bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_b1.err; echo "headline done" >&2
python3 bench.py --workload boxes_spectral --steps 1 --warmup 1 > gpurun_out/${tag}_bench_boxes_spectral.json 2> gpurun_out/${tag}_b2.err
python3 bench.py --workload ibl --steps 1 --warmup 1 > gpurun_out/${tag}_bench_ibl.json 2> gpurun_out/${tag}_b3.err; echo "configs 2,3 done" >&2
python3 bench.py --workload grid10m --steps 1 --warmup 0 > gpurun_out/${tag}_bench_grid10m.json 2> gpurun_out/${tag}_b4.err; echo "grid done" >&2
python3 bench.py --workload grid10m --instanced --steps 1 --warmup 0 > gpurun_out/${tag}_bench_grid10m_instanced.json 2> gpurun_out/${tag}_b5.err; echo "all done" >&2
