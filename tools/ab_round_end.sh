#!/usr/bin/env bash
# tools/ab_round_end.sh -- (GPU box) the A/B runs of round 4's last hour, rewritten per experiment (git log -p shows the earlier ones: one more
# wave per SIMD for three kernels; an even split of k_raster_small's items); results in profiles/r04_occupancy_ab.txt.  This version: the HIP
# runtime's kernel-argument placement.
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
