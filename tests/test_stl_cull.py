"""the ASCII-STL mesh input (rasteriser/Source/LoadSTL.cpp) and the cull step on the GPU.
The loader has no recorded reference output here (parity unpinned); the test files are written by the tests in the shape
of the reference's enemy1.stl ("solid AssimpScene", one-space indents, blank line between facets)."""
import os

import numpy as np
import pytest

import mirt
from mirt_oracle import DEFAULT_LIGHT


def write_stl(path, tris, eol="\n", extra_spaces=False):
    """tris: (n, 3, 3) vertices.  Formatted like the reference's asset; optionally with CRLF and doubled spaces."""
    sp = "  " if extra_spaces else " "
    with open(path, "w", newline="") as f:
        f.write("solid AssimpScene" + eol)
        for t in tris:
            f.write(" facet normal 0 0 0" + eol + "  outer loop" + eol)
            for v in t:
                f.write("  vertex" + sp + sp.join("%.6g" % c for c in v) + eol)
            f.write("  endloop" + eol + " endfacet" + eol + eol)
        f.write("endsolid AssimpScene" + eol)


def sphere(nu, nv, r=20.0):
    """A UV sphere in the unscaled units of the reference's asset (the loader multiplies by -0.05)."""
    tris = []
    for i in range(nu):
        for j in range(nv):
            def p(a, b):
                th, ph = np.pi * a / nu, 2 * np.pi * b / nv
                return (r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph))
            a, b, c, d = p(i, j), p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)
            tris.append((a, b, c))
            tris.append((a, c, d))
    return np.array(tris)


def test_oracle_loader_follows_the_reference_parser(oracle, tmp_path):
    """-scale on every coordinate, grey colour, normal = normalize(cross(e2, e1)); CRLF files and runs of spaces parse the
    same (split drops empty tokens, atof stops at the carriage return)."""
    tris = sphere(6, 8)
    a, b = str(tmp_path / "a.stl"), str(tmp_path / "b.stl")
    write_stl(a, tris)
    write_stl(b, tris, eol="\r\n", extra_spaces=True)
    ta, tb = oracle.load_stl(a), oracle.load_stl(b)
    assert ta.shape == (len(tris), 15) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32))
    parsed = np.array([[float("%.6g" % c) for c in v] for t in tris for v in t], np.float64).astype(np.float32).reshape(-1, 9)
    assert np.array_equal(ta[:, :9], parsed * np.float32(-0.05))
    assert np.all(ta[:, 12:] == np.float32(0.5))
    e1, e2 = ta[:, 3:6] - ta[:, 0:3], ta[:, 6:9] - ta[:, 0:3]
    nrm = np.cross(e2.astype(np.float64), e1.astype(np.float64))
    ok = np.linalg.norm(nrm, axis=1) > 0                    # the poles give zero-area triangles: normal = NaN, as in the reference
    assert np.allclose(ta[ok, 9:12], nrm[ok] / np.linalg.norm(nrm[ok], axis=1, keepdims=True), atol=1e-5)
    assert np.all(np.isnan(ta[~ok, 9:12]))
    with pytest.raises(ValueError):
        oracle.load_stl(str(tmp_path / "missing.stl"))
    broken = tmp_path / "broken.stl"
    broken.write_text("solid x\n facet normal 0 0 0\n  outer loop\n  vertex 1 2\n")
    with pytest.raises(ValueError):
        oracle.load_stl(str(broken))


def test_reference_asset_facet_count(oracle):
    """The reference's own mesh (read as data, where the reference tree is present): 9 028 facets."""
    path = "/root/reference/rasteriser/Source/enemy1.stl"
    if not os.path.exists(path):
        pytest.skip("reference tree not present")
    t = oracle.load_stl(path)
    assert t.shape == (9028, 15)
    assert np.isfinite(t[:, :9]).all() and np.abs(t[:, :9]).max() < 10.0


def test_product_loader_matches_oracle_without_a_gpu(oracle, tmp_path):
    """mirt_scene_load_stl is host code: bit-identical to the oracle's restatement, same error behaviour."""
    tris = sphere(9, 12)
    p = str(tmp_path / "s.stl")
    write_stl(p, tris, eol="\r\n")
    got = mirt.scene_load_stl(p)
    want = oracle.load_stl(p)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(mirt.scene_load_stl(p, 0.1, (0.2, 0.3, 0.4)).view(np.uint32), oracle.load_stl(p, 0.1, (0.2, 0.3, 0.4)).view(np.uint32))
    with pytest.raises(mirt.MirtError):
        mirt.scene_load_stl(str(tmp_path / "missing.stl"))


@pytest.fixture()
def device():
    mirt.init(0)
    yield
    mirt.shutdown()


@pytest.mark.gpu
def test_stl_mesh_renders_like_the_oracle(oracle, tmp_path, device):
    """A few thousand facets from an STL file through both renderers, with the reference's custom-model camera
    (cameraPos = (0, -0.5, -5), rasteriser.cpp:110): owner index, depth, colours and surface identical to the oracle."""
    p = str(tmp_path / "sphere.stl")
    write_stl(p, sphere(24, 40))
    tris = mirt.scene_load_stl(p)
    assert np.array_equal(tris.view(np.uint32), oracle.load_stl(p).view(np.uint32))
    tris = tris[np.isfinite(tris).all(axis=1)]              # zero-area pole facets carry NaN normals (as in the reference)
    W, H = 320, 240
    cam, rot = (0, -0.5, -5.0), oracle.rot_from_yaw(0.0, 1.01)
    view = mirt.make_view(cam, rot, float(H), W, H)
    mirt.scene_upload(tris)
    mirt.cull_device(view, 3)
    culled = mirt.scene_get_culled()
    assert np.array_equal(culled, oracle.cull(tris, cam, rot, float(H), W, H, 3))
    assert 0 < culled.sum() < len(tris)
    ref = oracle.rasterise(tris, culled, cam, rot, float(H), W, H, DEFAULT_LIGHT)
    got = mirt.rasterise(view, DEFAULT_LIGHT)
    assert np.array_equal(got["index"], ref["index"]) and np.array_equal(got["xrgb"], ref["xrgb"])
    assert np.array_equal(got["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    rref = oracle.raytrace(tris, cam, oracle.rot_from_yaw(0.0, 1.0), float(H), W, H, DEFAULT_LIGHT)
    rgot = mirt.raytrace(mirt.make_view(cam, oracle.rot_from_yaw(0.0, 1.0), float(H), W, H), DEFAULT_LIGHT)
    assert np.array_equal(rgot["index"], rref["index"]) and np.array_equal(rgot["xrgb"], rref["xrgb"])
    assert (rgot["index"] >= 0).sum() > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
@pytest.mark.parametrize("yaw,cam", [(0.0, (0, 0, -3)), (0.6, (0.4, -0.2, -2.5)), (-2.8, (0.1, 0.3, 1.0))])
def test_cull_on_the_device_matches_host_and_oracle(oracle, device, flags, yaw, cam):
    """mirt_cull_device == mirt_cull (host) == oracle, flag for flag, on the Cornell box and a soup around the camera."""
    for tris in (mirt.scene_cornell(), mirt.scene_soup(5, 20000, 0.3)):
        rot = oracle.rot_from_yaw(yaw, 1.01)
        view = mirt.make_view(cam, rot, 500.0, 500, 500)
        mirt.scene_upload(tris)
        mirt.cull_device(view, flags)
        got = mirt.scene_get_culled()
        assert np.array_equal(got, mirt.cull(tris, view, flags))
        assert np.array_equal(got, oracle.cull(tris, cam, rot, 500.0, 500, 500, flags))


@pytest.mark.gpu
def test_host_adapter_custom_model_build(oracle, tmp_path):
    """host/demo_main rasterstl = the reference's CUSTOM_MODEL main(): LoadSTLFile, cameraPos (0, -0.5, -5), Update() with
    the cull on the GPU, Draw().  Its surface must hold the words the oracle produces for the same mesh."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cpp-raytracer-rasterizer_amd", "host", "demo_main")
    assert os.path.exists(exe), "host/demo_main not built (make -C cpp-raytracer-rasterizer_amd)"
    stl = str(tmp_path / "model.stl")
    sph = sphere(20, 30)
    sph = sph[np.linalg.norm(np.cross(sph[:, 2] - sph[:, 0], sph[:, 1] - sph[:, 0]), axis=1) > 1e-6]   # no zero-area facets
    write_stl(stl, sph)
    W = H = 200
    raw, bmp = str(tmp_path / "out.xrgb"), str(tmp_path / "out.bmp")
    subprocess.run([exe, "rasterstl", str(W), str(H), bmp, raw, stl], check=True, timeout=300)
    got = np.fromfile(raw, np.uint32).reshape(H, W)
    tris = oracle.load_stl(stl)
    cam, rot = (0, -0.5, -5.0), oracle.rot_from_yaw(0.0, 1.01)
    culled = oracle.cull(tris, cam, rot, float(H), W, H, 3)
    ref = oracle.rasterise(tris, culled, cam, rot, float(H), W, H, DEFAULT_LIGHT)["xrgb"]
    assert np.array_equal(got, ref)
    assert (ref != 0).sum() > 500


@pytest.mark.gpu
def test_cull_on_the_device_is_ordered_between_frames_in_flight(oracle, device):
    """mirt_cull_device with two frames in flight: the kernel must wait for the frames of BOTH streams and the next frame
    -- which takes the other stream -- must wait for the kernel.  Frame A (view a), cull for view b, frame B: both frames
    must equal the oracle's, on a mesh large enough (120 k triangles) for the cull kernel to still run when frame B's
    vertex kernel starts."""
    from devbuf import DeviceArray
    tris = mirt.scene_soup(9, 120000, 0.03)
    W, H = 480, 270
    views = [((0.0, 0.0, -3.0), 0.0), ((0.9, 0.1, -2.4), 0.5)]
    mirt.scene_upload(tris)
    want = []
    for cam, yaw in views:
        rot = oracle.rot_from_yaw(yaw, 1.01)
        culled = oracle.cull(tris, cam, rot, float(H), W, H, 3)
        want.append(oracle.rasterise(tris, culled, cam, rot, float(H), W, H, DEFAULT_LIGHT, want=("xrgb", "index")))
    assert not np.array_equal(want[0]["index"], want[1]["index"])
    vs = [mirt.make_view(cam, oracle.rot_from_yaw(yaw, 1.01), float(H), W, H) for cam, yaw in views]
    surf = [DeviceArray((H, W), np.uint32, 0x11), DeviceArray((H, W), np.uint32, 0x22)]
    idx = [DeviceArray((H, W), np.int32, 0x33), DeviceArray((H, W), np.int32, 0x44)]
    mirt.set_frames_in_flight(2)
    try:
        for rep in range(6):
            for k in (0, 1):
                mirt.cull_device(vs[k], 3)
                mirt.rasterise_device(vs[k], DEFAULT_LIGHT, (0.2, 0.2, 0.2), 0, H, 0, surf[k].ptr, W * 4, None, None, idx[k].ptr)
        for k in (0, 1):
            assert np.array_equal(idx[k].read(), want[k]["index"]), "view %d: owner index differs" % k
            assert np.array_equal(surf[k].read(), want[k]["xrgb"]), "view %d: surface differs" % k
    finally:
        mirt.set_frames_in_flight(1)
        for b in surf + idx:
            b.free()


# ---- the cull flags' way through the streams: the sequences the call-sequence fuzzer caught in round 3, as deterministic tests ----
# ops: ("cull", view) = mirt_cull_device for that view; ("rt",) = a ray-traced frame (takes a stream's turn, reads no flags);
# ("raster", view) / ("band", view, part) = a rasterised frame / its upper (0) or lower (1) half.  Every rasterised frame must show
# the flags of the most recent cull step, whatever stream it lands on.
_CULL_SEQUENCES = {
    # cull -> ray-trace -> rasterise: the ray-traced frame takes the stream the cull step wrote the flags for
    "cull_raytrace_rasterise": [op for k in range(8) for op in (("cull", k % 2), ("rt",), ("raster", k % 2))],
    # frames drawn WITHOUT a cull step reuse the latest flags across streams
    "frames_without_a_cull_step": [("cull", 0), ("raster", 0), ("raster", 1), ("raster", 0), ("rt",), ("raster", 1), ("cull", 1), ("raster", 1),
                                   ("raster", 0), ("rt",), ("rt",), ("raster", 0), ("raster", 1), ("cull", 0), ("rt",), ("raster", 1), ("raster", 0)],
    # a stream reads two different copies in turn, then a cull step goes into the first of them
    "two_copies_then_a_cull_into_the_first": [("cull", 0), ("rt",), ("raster", 0), ("cull", 1), ("rt",), ("rt",), ("raster", 1), ("cull", 0), ("raster", 0),
                                              ("rt",), ("raster", 1), ("cull", 1), ("rt",), ("raster", 0), ("raster", 1)] * 2,
    # two bands per frame, a cull step per frame: with three frames in flight the second band's hand-over copy overwrites a copy
    # another stream may still be reading (advisor finding of round 3: raster_enqueue did not wait for such readers)
    "two_bands_per_frame": [op for k in range(9) for op in (("cull", k % 2), ("band", k % 2, 0), ("band", k % 2, 1))],
    # ... and a cull step every other frame only
    "two_bands_cull_every_other_frame": [op for k in range(10) for op in ((("cull", (k // 2) % 2),) if k % 2 == 0 else ()) + (("band", k % 2, 0), ("band", k % 2, 1))],
}


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight", [2, 3, 4])
@pytest.mark.parametrize("name", sorted(_CULL_SEQUENCES))
def test_cull_flags_follow_the_frames_through_the_streams(oracle, device, name, in_flight):
    """Cross-stream hand-over of the cull flags (mirt_cull_device / raster_enqueue) on a mesh large enough (120 k triangles) for the
    kernels of consecutive calls to overlap; every rasterised frame against the oracle's frame for (its view, the flags of the most
    recent cull step)."""
    from devbuf import DeviceArray
    tris = mirt.scene_soup(9, 120000, 0.03)
    W, H = 320, 180
    cams = [((0.0, 0.0, -3.0), 0.0), ((0.9, 0.1, -2.4), 0.5)]
    rots = [oracle.rot_from_yaw(yaw, 1.01) for _, yaw in cams]
    vs = [mirt.make_view(cam, rots[i], float(H), W, H) for i, (cam, _) in enumerate(cams)]
    culled = [oracle.cull(tris, cams[j][0], rots[j], float(H), W, H, 3) for j in range(2)]
    assert not np.array_equal(culled[0], culled[1])
    cache = {}

    def want(i, j):                                         # view i drawn with the flags view j's cull step left
        if (i, j) not in cache:
            cache[(i, j)] = oracle.rasterise(tris, culled[j], cams[i][0], rots[i], float(H), W, H, DEFAULT_LIGHT, want=("xrgb",))["xrgb"]
        return cache[(i, j)]

    mirt.scene_upload(tris)
    mirt.set_frames_in_flight(in_flight)
    scratch = DeviceArray((H, W), np.uint32, 0x5A)
    outs, checks = [], []
    try:
        flags = None
        for op in _CULL_SEQUENCES[name]:
            if op[0] == "cull":
                mirt.cull_device(vs[op[1]], 3)
                flags = op[1]
            elif op[0] == "rt":
                mirt.raytrace_device(vs[0], DEFAULT_LIGHT, (0.2, 0.2, 0.2), mirt.RT_AUTO, 0, H, 0, scratch.ptr, W * 4)
            else:
                y0, y1 = (0, H) if op[0] == "raster" else ((0, H // 2) if op[2] == 0 else (H // 2, H))
                buf = DeviceArray((H, W), np.uint32, 0x5A)
                outs.append(buf)
                mirt.rasterise_device(vs[op[1]], DEFAULT_LIGHT, (0.2, 0.2, 0.2), y0, y1, 0, buf.ptr, W * 4)
                checks.append((buf, op[1], flags, y0, y1))
        mirt.sync()
        for n, (buf, i, j, y0, y1) in enumerate(checks):
            got = buf.read()
            assert np.array_equal(got[y0:y1], want(i, j)[y0:y1]), "rasterised call %d (view %d, flags of view %d, rows %d..%d): %d words differ" % (
                n, i, j, y0, y1, int((got[y0:y1] != want(i, j)[y0:y1]).sum()))
    finally:
        mirt.set_frames_in_flight(1)
        for b in outs + [scratch]:
            b.free()
