#!/usr/bin/env bash
# tools/build_variant.sh <name> <source.hip> <extra hipcc flags...> -- an A/B build of the library: ONE translation unit compiled with
# extra flags, linked with the regular objects into cpp-raytracer-rasterizer_amd/build/variants/libmirt_<name>.so (git-ignored; travels to the GPU
# box).  tools/soup_variant.py <that .so> runs it.
set -euo pipefail
name="$1"; src="$2"; shift 2
cd "$(dirname "$0")/../cpp-raytracer-rasterizer_amd"
make -s -j8 libmirt.so
mkdir -p build/variants
obj="build/variants/${name}_$(basename "$src").o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fvisibility=hidden -Wall -Wno-unused-function "$@" -x hip -c "csrc/$src" -o "$obj"
objs=""
for o in build/*.o; do
    if [ "$(basename "$o")" = "$(basename "$src").o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "build/variants/libmirt_${name}.so" -ldl -Wl,-rpath,/opt/rocm/lib
echo "built build/variants/libmirt_${name}.so"
