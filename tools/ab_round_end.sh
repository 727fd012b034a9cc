set -uo pipefail
V=cpp-raytracer-rasterizer_amd/build/variants
{
echo "== raster4k: base, small6 (6 WGs per CU), base with 6 WGs per CU"
python tools/frame_variant.py raster4k
MIRT_SMALL_WGS_PER_CU=6 python tools/frame_variant.py raster4k $V/libmirt_small6.so
echo "== cornell1080: base, tile7"
python tools/frame_variant.py cornell1080 $V/libmirt_tile7.so
echo "== soup100k moving camera: base, scat5"
python tools/moving_ab.py soup100k $V/libmirt_scat5.so
echo "== soup1m8k moving camera: base, scat5"
python tools/moving_ab.py soup1m8k $V/libmirt_scat5.so
} > gpurun_out/ab1.txt 2>&1
tail -30 gpurun_out/ab1.txt
