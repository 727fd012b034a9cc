set -uo pipefail
V=cpp-raytracer-rasterizer_amd/build/variants
{
echo "== raster4k: base; even split of k_raster_small's items over the waves, 5 / 6 / 8 workgroups per CU; base with 4"
python tools/frame_variant.py raster4k $V/libmirt_even.so
MIRT_SMALL_WGS_PER_CU=6 python tools/frame_variant.py raster4k $V/libmirt_even.so
MIRT_SMALL_WGS_PER_CU=8 python tools/frame_variant.py raster4k $V/libmirt_even.so
MIRT_SMALL_WGS_PER_CU=4 python tools/frame_variant.py raster4k $V/libmirt_even.so
} > gpurun_out/ab2.txt 2>&1
cat gpurun_out/ab2.txt
