#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "EXTRA FLAGS" — an alternate libslrhip with one translation unit rebuilt with extra
# -D flags, for A/B timing on the GPU box: SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/../slr_amd/csrc"
name=$1; src=$2; extra=$3
mkdir -p variants
make -s libslrhip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $extra -c $src -o variants/${src%.hip}_$name.o
objs=""
for o in pt_trace.o pt_trace_ws.o pt_trace_quad.o pt_shade.o pt_shade_rgb.o pt_shade_spec16.o pt_shade_specq.o pt_shade_multi.o pt_shade_multi_rgb.o pt_shade_multi_spec.o pt_shade_tex_rgb.o pt_shade_tex_spec.o pt_tail.o pt_tail_rgb.o pt_tail_spec16.o pt_tail_multi_rgb.o pt_tail_tex_rgb.o slrhip_api.o bvh.o sbvh.o host_util.o; do
  if [ "$o" = "${src%.hip}.o" ]; then objs="$objs variants/${src%.hip}_$name.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -o variants/libslrhip_$name.so $objs
echo variants/libslrhip_$name.so
