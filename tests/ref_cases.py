"""The frame configurations both pins share: tests/test_oracle_ref_render.py runs them through the reference's own text
(oracle/_ref, in the build container) against the CPU restatement, and tests/golden/make_golden.py records the reference's
outputs for them as hashes (tests/golden/ref_render.json), which the restatement AND the GPU path must reproduce wherever
the reference is absent.  Scenes are built from generators that live in this repository (Cornell box, mt19937 soups), so
only the expected outputs come from the reference.

Frames are the sizes the reference itself can render: 500x500, and 150x150 (its -DREALTIME build) for the ray tracer.
"""
import numpy as np

L0 = [0.0, -0.5, -0.7, 1.0, 1.0, 1.0, 14.0]                 # AddLight(vec3(0,-0.5f,-0.7f), vec3(1,1,1), 14), raytracer.cpp:116
L1 = [0.5, 0.3, -0.9, 1.0, 0.5, 0.2, 6.0]
L2 = [-0.6, -0.2, 0.1, 0.3, 0.9, 0.4, 9.0]

# name: size, scene, camera position, yaw, focal length, lights, aa, soft-shadow samples
RT_CASES = {
    "cornell500_default": dict(size=500, scene=("cornell",), cam=(0, 0, -2), yaw=0.0, focal=250.0, lights=[L0], aa=1, soft=1),
    "cornell500_nolight": dict(size=500, scene=("cornell",), cam=(0, 0, -2), yaw=0.0, focal=250.0, lights=[], aa=1, soft=1),
    "cornell500_yaw_moved_3lights": dict(size=500, scene=("cornell",), cam=(0.2, -0.1, -2.5), yaw=0.3, focal=300.0, lights=[L0, L1, L2], aa=1, soft=1),
    "cornell_soup400_500_yaw": dict(size=500, scene=("cornell+soup", 7, 400, 0.12), cam=(0.1, 0.05, -2.2), yaw=-0.25, focal=250.0, lights=[L0, L1], aa=1, soft=1),
    "cornell150_realtime_default": dict(size=150, scene=("cornell",), cam=(0, 0, -4.3), yaw=0.0, focal=250.0, lights=[L0], aa=1, soft=1),
    "soup2000_150_inside": dict(size=150, scene=("soup", 3, 2000, 0.1), cam=(0.1, -0.2, -0.3), yaw=0.7, focal=75.0, lights=[L0, L2], aa=1, soft=1),
    "cornell150_soft16": dict(size=150, scene=("cornell",), cam=(0, 0, -2), yaw=0.0, focal=75.0, lights=[L0], aa=1, soft=16),
    "cornell150_aa3": dict(size=150, scene=("cornell",), cam=(0, 0, -2), yaw=0.1, focal=75.0, lights=[L0], aa=3, soft=1),
    "cornell_soup150_aa2_soft4_2lights": dict(size=150, scene=("cornell+soup", 5, 60, 0.2), cam=(0, 0.1, -2.1), yaw=-0.2, focal=80.0, lights=[L0, L1], aa=2, soft=4),
}

# name: scene, camera position, yaw, focal length, cameraRot[1][1], lights, cull flags (bit0 back face, bit1 frustum), FOCAL_LENGTH
RASTER_CASES = {
    "cornell_default": dict(scene=("cornell",), cam=(0, 0, -3), yaw=0.0, focal=500.0, rot11=1.01, lights=[L0], flags=3, focal_plane=1.9),
    "cornell_yaw_offscreen": dict(scene=("cornell",), cam=(0.9, 0.1, -2.4), yaw=0.5, focal=500.0, rot11=1.01, lights=[L0], flags=3, focal_plane=1.9),
    "cornell_close_offscreen_2lights": dict(scene=("cornell",), cam=(-0.3, 0.2, -1.9), yaw=-0.35, focal=420.0, rot11=1.01, lights=[L0, L1], flags=1, focal_plane=1.3),
    "cornell_soup300_nocull": dict(scene=("cornell+soup", 9, 300, 0.2), cam=(0.1, 0, -3), yaw=0.1, focal=500.0, rot11=1.01, lights=[L0], flags=0, focal_plane=1.9),
    "soup4000_frustum_only": dict(scene=("soup", 4, 4000, 0.08), cam=(0.3, 0.1, -2.6), yaw=0.4, focal=500.0, rot11=1.0, lights=[L0, L2], flags=2, focal_plane=2.5),
}


def build_scene(oracle, spec):
    if spec[0] == "cornell":
        return oracle.cornell()
    soup = oracle.soup(spec[1], spec[2], spec[3])
    return soup if spec[0] == "soup" else np.concatenate([oracle.cornell(), soup])


def lights_array(lights):
    return np.array(lights, np.float32).reshape(-1, 7)
