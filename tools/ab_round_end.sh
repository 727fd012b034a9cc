#!/usr/bin/env bash
# tools/ab_round_end.sh -- (GPU box) the A/B runs of round 4's last hour, rewritten per experiment (git log -p shows the earlier ones: one more
# wave per SIMD for three kernels; an even split of k_raster_small's items; the runtime's kernel-argument placement); results in
# profiles/r04_occupancy_ab.txt.  This version: the backend's max-ILP scheduling strategy for the trace and the tile kernel
# (tools/build_variant.sh ilptrace rt_trace.hip -mllvm -amdgpu-sched-strategy=max-ilp; the same for rt_tile.hip).
set -uo pipefail
V=cpp-raytracer-rasterizer_amd/build/variants
{
echo "== soup100k moving camera: base, max-ilp trace kernel"
python tools/moving_ab.py soup100k $V/libmirt_ilptrace.so
echo "== soup1m8k"
python tools/moving_ab.py soup1m8k $V/libmirt_ilptrace.so
echo "== cornell1080: base, max-ilp tile kernel"
python tools/frame_variant.py cornell1080 $V/libmirt_ilptile.so
} > gpurun_out/ab4.txt 2>&1
cat gpurun_out/ab4.txt
