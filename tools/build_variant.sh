#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>"  ->  build_variants/libhrcore_NAME.so  (A/B experiments; load with HRCORE_LIB)
set -e
name="$1"; extra="$2"
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp="$(mktemp -d)"; mkdir -p "$root/build_variants"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-function -Wno-unused-result -Wno-unused-value -I$root/include $extra"
for f in hr_core hr_render hr_build; do
  /opt/rocm/bin/hipcc $FLAGS -c "$root/heatray_amd/csrc/$f.hip" -o "$tmp/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/build_variants/libhrcore_$name.so" "$tmp"/*.o
rm -rf "$tmp"; echo "built build_variants/libhrcore_$name.so"
