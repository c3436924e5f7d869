#!/bin/bash
# tools/build_variant.sh NAME "EXTRA FLAGS" FILE.hip [FILE.hip ...] — an alternate libslrhip with the named translation units
# rebuilt with extra -D flags (and -DSLR_TUNING_KNOBS: the measurement environment variables are read), for A/B timing on the GPU box:
#     SLRHIP_LIBRARY=slr_amd/csrc/variants/libslrhip_NAME.so python bench.py ...
set -e
cd "$(dirname "$0")/../slr_amd/csrc"
name=$1; extra=$2; shift 2
mkdir -p variants/$name
make -s libslrhip.so
objs=$(make -s --eval='print-objs: ; @echo $(OBJS)' print-objs)
link=""
for o in $objs; do
  src=${o%.o}.hip
  rebuilt=0
  for f in "$@"; do if [ "$f" = "$src" ]; then rebuilt=1; fi; done
  if [ $rebuilt = 1 ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fno-slp-vectorize -DSLR_TUNING_KNOBS $extra -c $src -o variants/$name/$o &
    link="$link variants/$name/$o"
  else link="$link $o"; fi
done
wait
hipcc --offload-arch=gfx950 -shared -o variants/libslrhip_$name.so $link
echo variants/libslrhip_$name.so
