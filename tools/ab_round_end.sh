set -uo pipefail
{
for v in 0 1; do
echo "== HIP_FORCE_DEV_KERNARG=$v"
HIP_FORCE_DEV_KERNARG=$v python tools/moving_ab.py soup100k
HIP_FORCE_DEV_KERNARG=$v python tools/frame_variant.py raster4k
HIP_FORCE_DEV_KERNARG=$v python tools/frame_variant.py cornell1080
done
} > gpurun_out/ab3.txt 2>&1
cat gpurun_out/ab3.txt
