"""BASELINE.json configs 3 and 5 at their STATED size, on the GPU, inside the driver's `pytest -m gpu` run.

    config 3: 100 000 random triangles (seed 1, s 0.05), 1920x1080
    config 5: 1 000 000 random triangles (seed 2, s 0.02), 7680x4320, eight 540-row bands

Brute force over the whole frame is out of the CPU oracle's reach at these sizes (3e11 and 6.6e13 ray-triangle tests), so
the oracle (`ClosestIntersection` / `DirectLight` restated, raytracer.cpp:202-257, 265-327) checks spread single rows --
closest-hit index, float colour bits and XRGB words -- and size-independent properties cover the rest of the frame:
binned == brute force byte for byte over the full frame (config 3), and bands rendered one by one == the full frame byte
for byte (config 5; what each of the eight GPUs renders when the frame is sharded).
"""
import numpy as np
import pytest

import mirt
from devbuf import DeviceArray
from mirt_oracle import DEFAULT_LIGHT

pytestmark = pytest.mark.gpu
INDIRECT = (0.2, 0.2, 0.2)


@pytest.fixture(scope="module", autouse=True)
def device():
    mirt.init(0)
    yield
    mirt.shutdown()


def _planes(W, H):
    return DeviceArray((H, W), np.uint32, 0x5A), DeviceArray((H, W, 3), np.float32, 0x5A), DeviceArray((H, W), np.int32, 0x5A)


def _render(view, mode, W, H, bands=None):
    """One frame into fresh device planes, as one call or band by band; returns (xrgb, rgb bits, index) on the host."""
    x, c, i = _planes(W, H)
    try:
        for (y0, y1) in (bands or [(0, H)]):
            mirt.raytrace_device(view, DEFAULT_LIGHT, INDIRECT, mode, y0, y1, 0, x.ptr, W * 4, c.ptr, i.ptr)
        return x.read(), c.read().view(np.uint32), i.read()
    finally:
        for b in (x, c, i):
            b.free()


def _check_rows_against_oracle(oracle, tris, cam, rot, focal, W, H, rows, got, threads=16):
    gx, gc, gi = got
    for y in rows:
        ref = oracle.raytrace(tris, cam, rot, focal, W, H, DEFAULT_LIGHT, y0=y, y1=y + 1, threads=threads, want=("rgb", "index", "xrgb"))
        assert np.array_equal(gi[y], ref["index"][y]), "row %d: closest-hit index differs in %d pixels" % (y, int((gi[y] != ref["index"][y]).sum()))
        assert np.array_equal(gc[y], ref["rgb"][y].view(np.uint32)), "row %d: float colours not bit-identical" % y
        if 1 <= y < H - 1:
            assert np.array_equal(gx[y, 1:-1], ref["xrgb"][y, 1:-1]), "row %d: XRGB words differ" % y


def test_config3_soup100k_1080p_full_size(oracle):
    """Config 3: binned == brute force over the whole 1920x1080 frame (3e11 tests on the GPU), 16 spread rows of it == the
    oracle, and the same rows rendered as single-row bands == the full frame."""
    W, H, cam, focal = 1920, 1080, (0, 0, -2), 540.0
    rot = oracle.rot_from_yaw(0.0, 1.0)
    tris = mirt.scene_soup(1, 100000, 0.05)
    mirt.scene_upload(tris)
    view = mirt.make_view(cam, rot, focal, W, H)
    binned = _render(view, mirt.RT_BINNED, W, H)
    assert mirt.stats()["mode_used"] == mirt.RT_BINNED
    shadow_binned = mirt.stats()["shadow_rays"]
    brute = _render(view, mirt.RT_BRUTE, W, H)
    assert mirt.stats()["mode_used"] == mirt.RT_BRUTE and mirt.stats()["shadow_rays"] == shadow_binned
    # the border words are never written by the ray tracer (raytracer.cpp:618-620): both frames keep the fill there
    for a, b, what in zip(binned, brute, ("XRGB words", "float colours", "closest-hit index")):
        assert np.array_equal(a, b), "binned != brute force: %s differ in %d places" % (what, int((a != b).sum()))
    assert int((binned[2] >= 0).sum()) == shadow_binned            # one light: shadow rays = pixels whose primary ray hit
    rows = [0, 1, 71, 143, 215, 287, 359, 431, 503, 540, 575, 647, 719, 863, 1007, 1079]
    _check_rows_against_oracle(oracle, tris, cam, rot, focal, W, H, rows, binned)
    single = _render(view, mirt.RT_BINNED, W, H, bands=[(y, y + 1) for y in rows])
    for y in rows:
        for a, b in zip(single, binned):
            assert np.array_equal(a[y], b[y]), "single-row band %d differs from the full frame" % y


def test_config5_soup1m_8k_eight_bands(oracle):
    """Config 5: the eight 540-row bands of the 7680x4320 frame rendered one by one == the frame rendered at once, byte
    for byte (XRGB, float colours, index), and three rows of it == the oracle (1.5e10 tests each on the host)."""
    W, H, cam, focal = 7680, 4320, (0, 0, -2), 2160.0
    rot = oracle.rot_from_yaw(0.0, 1.0)
    tris = mirt.scene_soup(2, 1000000, 0.02)
    mirt.scene_upload(tris)
    view = mirt.make_view(cam, rot, focal, W, H)
    full = _render(view, mirt.RT_AUTO, W, H)
    assert mirt.stats()["mode_used"] == mirt.RT_BINNED
    banded = _render(view, mirt.RT_AUTO, W, H, bands=[(k * 540, (k + 1) * 540) for k in range(8)])
    for a, b, what in zip(banded, full, ("XRGB words", "float colours", "closest-hit index")):
        assert np.array_equal(a, b), "bands != full frame: %s differ in %d places" % (what, int((a != b).sum()))
    hit = int((full[2] >= 0).sum())
    assert 0.2 * W * H < hit < W * H                                # a soup, not an empty or a solid frame
    _check_rows_against_oracle(oracle, tris, cam, rot, focal, W, H, [539, 2160, 3781], full)
