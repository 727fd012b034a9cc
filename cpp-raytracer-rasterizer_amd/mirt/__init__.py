"""mirt -- thin ctypes binding of the C-ABI in include/mirt.h (libmirt.so, hand-written gfx950 kernels).

This module is plumbing for tests and bench.py: it marshals numpy arrays (or device pointers of torch
tensors) into the C entry points.  There is no Python/CPU implementation of the render path here and no
fallback: if libmirt.so is missing or no MI355X is visible, the calls raise MirtError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libmirt.so")

MAX_LIGHTS = 32
RT_AUTO, RT_BRUTE, RT_BINNED = 0, 1, 2
KERNEL_NAMES = ("prep", "bin", "trace", "dof", "raster_setup", "raster_frag", "raster_resolve", "clear")

# every symbol include/mirt.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = (
    "mirt_init", "mirt_shutdown", "mirt_last_error", "mirt_abi_version", "mirt_set_profiling", "mirt_sync",
    "mirt_stream", "mirt_scene_upload", "mirt_scene_set_culled", "mirt_scene_size", "mirt_scene_cornell",
    "mirt_scene_soup", "mirt_scene_load_stl", "mirt_cull", "mirt_cull_device", "mirt_scene_get_culled", "mirt_set_soft_shadows", "mirt_set_antialiasing", "mirt_set_depth_of_field", "mirt_set_frames_in_flight", "mirt_raytrace", "mirt_raytrace_device", "mirt_raytrace_ex", "mirt_raytrace_device_ex", "mirt_rasterise",
    "mirt_rasterise_device", "mirt_get_stats", "mirt_get_previous_kernel_ms", "mirt_surface_register", "mirt_surface_unregister", "mirt_raytrace_async", "mirt_rasterise_async",
    "mirt_band_of", "mirt_band_plan", "mirt_set_partition", "mirt_partition_segments", "mirt_partition_plan",
    "mirt_set_cost_histogram", "mirt_cost_histogram", "mirt_weighted_bounds", "mirt_partition_bounds", "mirt_bounds_plan", "mirt_comm_create_id", "mirt_comm_init", "mirt_comm_shutdown", "mirt_comm_selfcheck", "mirt_raytrace_sharded", "mirt_rasterise_sharded",
)


class MirtError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("rot", C.c_float * 9), ("focal", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("color", C.c_float * 3), ("intensity", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("tests", C.c_uint64),
                ("gpu_ms", C.c_float), ("kernel_ms", C.c_float * 8), ("mode_used", C.c_int32), ("candidates", C.c_uint64),
                ("steps_primary", C.c_uint64), ("steps_shadow", C.c_uint64), ("drains", C.c_uint64),
                ("bins_reused", C.c_uint32), ("selected_triangles", C.c_uint32)]


_vp = C.c_void_p
_lib = None


def load():
    """Loads libmirt.so (raises MirtError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirtError("libmirt.so not built: run `make -C cpp-raytracer-rasterizer_amd` (or __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    lib.mirt_last_error.restype = C.c_char_p
    lib.mirt_stream.restype = _vp
    lib.mirt_init.argtypes = [C.c_int]
    lib.mirt_set_profiling.argtypes = [C.c_int]
    lib.mirt_scene_upload.argtypes = [_vp, _vp, C.c_int]
    lib.mirt_scene_set_culled.argtypes = [_vp, C.c_int]
    lib.mirt_scene_cornell.argtypes = [_vp]
    lib.mirt_scene_soup.argtypes = [C.c_uint32, C.c_int, C.c_float, _vp]
    lib.mirt_cull.argtypes = [_vp, C.c_int, C.POINTER(View), C.c_int, _vp]
    lib.mirt_cull_device.argtypes = [C.POINTER(View), C.c_int]
    lib.mirt_scene_get_culled.argtypes = [_vp, C.c_int]
    lib.mirt_scene_load_stl.argtypes = [C.c_char_p, C.c_float, _vp, _vp, C.c_int]
    lib.mirt_set_soft_shadows.argtypes = [C.c_int, _vp, C.c_int]
    lib.mirt_set_depth_of_field.argtypes = [C.c_int, C.c_float]
    lib.mirt_set_frames_in_flight.argtypes = [C.c_int]
    lib.mirt_raytrace.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]
    lib.mirt_raytrace_device.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _vp, C.c_int, _vp, _vp]
    lib.mirt_raytrace_ex.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]
    lib.mirt_raytrace_device_ex.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                            _vp, C.c_int, _vp, _vp, _vp, _vp]
    lib.mirt_rasterise.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp]
    lib.mirt_rasterise_device.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int,
                                          _vp, C.c_int, _vp, _vp, _vp]
    lib.mirt_get_stats.argtypes = [C.POINTER(Stats)]
    lib.mirt_get_previous_kernel_ms.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.mirt_surface_register.argtypes = [_vp, C.c_size_t]
    lib.mirt_surface_unregister.argtypes = [_vp]
    lib.mirt_raytrace_async.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, C.c_int, _vp, C.c_int]
    lib.mirt_rasterise_async.argtypes = [C.POINTER(View), _vp, C.c_int, _vp, _vp, C.c_int]
    lib.mirt_band_of.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mirt_band_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]
    lib.mirt_set_partition.argtypes = [C.c_int]
    lib.mirt_partition_segments.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int]
    lib.mirt_partition_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]
    lib.mirt_set_cost_histogram.argtypes = [C.c_int]
    lib.mirt_cost_histogram.argtypes = [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mirt_weighted_bounds.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]
    lib.mirt_partition_bounds.argtypes = [C.c_int, C.c_int, C.c_int, _vp]
    lib.mirt_bounds_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int]
    lib.mirt_comm_create_id.argtypes = [_vp]
    lib.mirt_comm_init.argtypes = [_vp, C.c_int, C.c_int]
    lib.mirt_comm_selfcheck.argtypes = [C.c_size_t]
    lib.mirt_raytrace_sharded.argtypes = [C.POINTER(View), C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, C.c_int]
    lib.mirt_rasterise_sharded.argtypes = [C.POINTER(View), C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise MirtError("mirt error %d: %s" % (rc, load().mirt_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


def make_view(pos, rot9, focal, W, H):
    v = View()
    v.pos[:] = [float(x) for x in pos]
    v.rot[:] = [float(x) for x in np.asarray(rot9, np.float32).ravel()]
    v.focal = float(focal)
    v.width, v.height = int(W), int(H)
    return v


def make_lights(lights7):
    """lights7: (k, 7) array {pos, color, intensity} -> (ctypes array or None, k)."""
    l = np.ascontiguousarray(lights7, np.float32).reshape(-1, 7)
    if len(l) == 0:
        return None, 0
    arr = (Light * len(l))()
    for i, r in enumerate(l):
        arr[i].pos[:] = r[0:3].tolist()
        arr[i].color[:] = r[3:6].tolist()
        arr[i].intensity = float(r[6])
    return arr, len(l)


def rot_from_yaw(yaw, m11):
    """cameraRot as the reference's Update() builds it (raytracer.cpp:377-382 / rasteriser.cpp:378-383)."""
    c, s = np.float32(np.cos(np.float32(yaw))), np.float32(np.sin(np.float32(yaw)))
    r = np.zeros(9, np.float32)
    r[4] = m11
    r[0], r[2], r[6], r[8] = c, s, -s, c
    return r


# ---- lifetime ---------------------------------------------------------------------------------------

def init(device=0):
    _check(load().mirt_init(int(device)))


def shutdown():
    if _lib is not None:
        _lib.mirt_shutdown()


def surface_register(arr):
    """Pins and maps a host surface (numpy array): frames rendered into it skip the staging copy (mirt_surface_register)."""
    _check(load().mirt_surface_register(_ptr(arr), arr.nbytes))


def surface_unregister(arr):
    _check(load().mirt_surface_unregister(_ptr(arr)))


def previous_kernel_ms():
    """kernel_ms (dict like stats()["kernel_ms"]) and gpu_ms of the call before the last one (two frames in flight, profiling on)."""
    k = (C.c_float * 8)()
    gms = C.c_float()
    _check(load().mirt_get_previous_kernel_ms(k, C.byref(gms)))
    return {name: float(k[i]) for i, name in enumerate(KERNEL_NAMES)}, float(gms.value)


def set_profiling(on):
    _check(load().mirt_set_profiling(1 if on else 0))


def sync():
    _check(load().mirt_sync())


def stats():
    s = Stats()
    _check(load().mirt_get_stats(C.byref(s)))
    return {"primary_rays": s.primary_rays, "shadow_rays": s.shadow_rays, "tests": s.tests, "gpu_ms": s.gpu_ms,
            "kernel_ms": dict(zip(KERNEL_NAMES, list(s.kernel_ms))), "mode_used": s.mode_used, "candidates": s.candidates,
            "steps_primary": s.steps_primary, "steps_shadow": s.steps_shadow, "drains": s.drains,
            "bins_reused": int(s.bins_reused), "selected_triangles": int(s.selected_triangles)}


# ---- scene ------------------------------------------------------------------------------------------

def scene_cornell():
    t = np.zeros((30, 15), np.float32)
    n = load().mirt_scene_cornell(_ptr(t))
    if n != 30:
        _check(n if n < 0 else -3)
    return t


def scene_soup(seed, n, s):
    t = np.zeros((n, 15), np.float32)
    rc = load().mirt_scene_soup(int(seed), int(n), float(s), _ptr(t))
    if rc != n:
        _check(rc if rc < 0 else -3)
    return t


def cull(tris, view, flags=3):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
    c = np.zeros(len(tris), np.uint8)
    _check(load().mirt_cull(_ptr(tris), len(tris), C.byref(view), int(flags), _ptr(c)))
    return c


def cull_device(view, flags=3):
    """The cull step on the GPU for the uploaded scene (flags stay on the device)."""
    _check(load().mirt_cull_device(C.byref(view), int(flags)))


def scene_get_culled():
    c = np.zeros(load().mirt_scene_size(), np.uint8)
    _check(load().mirt_scene_get_culled(_ptr(c), len(c)))
    return c


def scene_load_stl(path, scale=0.05, colour=(0.5, 0.5, 0.5)):
    """LoadSTL::LoadSTLFile: (n, 15) triangles of an ASCII STL, scaled by -scale, one colour, normals recomputed."""
    col = np.asarray(colour, np.float32)
    n = load().mirt_scene_load_stl(str(path).encode(), float(scale), _ptr(col), None, 0)
    if n < 0:
        raise MirtError("cannot load %s (status %d)" % (path, n))
    t = np.zeros((n, 15), np.float32)
    if n:
        _check(min(0, load().mirt_scene_load_stl(str(path).encode(), float(scale), _ptr(col), _ptr(t), n)))
    return t


def scene_upload(tris, culled=None):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
    culled = None if culled is None else np.ascontiguousarray(culled, np.uint8)
    _check(load().mirt_scene_upload(_ptr(tris), _ptr(culled), len(tris)))


def scene_set_culled(culled):
    culled = np.ascontiguousarray(culled, np.uint8)
    _check(load().mirt_scene_set_culled(_ptr(culled), len(culled)))


def set_depth_of_field(kernel_size, focal_length=0.0):
    """kernel_size x kernel_size blur before the pixels are stored (the reference: 8, FOCAL_LENGTH 1.3 / 1.9); <= 1: off."""
    _check(load().mirt_set_depth_of_field(int(kernel_size), float(focal_length)))


def set_frames_in_flight(frames):
    """1: device calls run in order on one stream; 2: they alternate between two streams (consecutive frames need
    different output planes)."""
    _check(load().mirt_set_frames_in_flight(int(frames)))


def set_antialiasing(samples):
    """samples x samples sub-rays per pixel (the reference's AA_SAMPLES = 3); <= 1 switches it off."""
    _check(load().mirt_set_antialiasing(int(samples)))


def set_soft_shadows(samples, positions=None):
    """samples <= 1 turns soft shadows off; otherwise positions is (nlights*samples, 3)."""
    if samples <= 1:
        _check(load().mirt_set_soft_shadows(1, None, 0))
        return
    p = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    _check(load().mirt_set_soft_shadows(int(samples), _ptr(p), len(p)))


# ---- render (host buffers) --------------------------------------------------------------------------

def raytrace(view, lights7, indirect=(0.2, 0.2, 0.2), mode=RT_AUTO, want_rgb=True, want_index=True, xrgb=None,
             want_intersection=False):
    """want_intersection adds "dist" and "pos": closestIntersections[].distance / .position (mirt_raytrace_ex)."""
    W, H = view.width, view.height
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32)
    out = {
        "xrgb": np.zeros((H, W), np.uint32) if xrgb is None else xrgb,
        "rgb": np.zeros((H, W, 3), np.float32) if want_rgb else None,
        "index": np.zeros((H, W), np.int32) if want_index else None,
    }
    if want_intersection:
        out["dist"] = np.zeros((H, W), np.float32)
        out["pos"] = np.zeros((H, W, 3), np.float32)
        _check(load().mirt_raytrace_ex(C.byref(view), larr, nl, _ptr(ind), int(mode), _ptr(out["xrgb"]),
                                       out["xrgb"].strides[0], _ptr(out["rgb"]), _ptr(out["index"]), _ptr(out["dist"]), _ptr(out["pos"])))
    else:
        _check(load().mirt_raytrace(C.byref(view), larr, nl, _ptr(ind), int(mode), _ptr(out["xrgb"]),
                                    out["xrgb"].strides[0], _ptr(out["rgb"]), _ptr(out["index"])))
    out["stats"] = stats()
    return out


def rasterise(view, lights7, indirect=(0.2, 0.2, 0.2), want_rgb=True, want_zinv=True, want_index=True, xrgb=None):
    W, H = view.width, view.height
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32)
    out = {
        "xrgb": np.full((H, W), 0xDEADBEEF, np.uint32) if xrgb is None else xrgb,
        "rgb": np.zeros((H, W, 3), np.float32) if want_rgb else None,
        "depth": np.zeros((H, W), np.float32) if want_zinv else None,
        "index": np.zeros((H, W), np.int32) if want_index else None,
    }
    _check(load().mirt_rasterise(C.byref(view), larr, nl, _ptr(ind), _ptr(out["xrgb"]), out["xrgb"].strides[0],
                                 _ptr(out["rgb"]), _ptr(out["depth"]), _ptr(out["index"])))
    out["stats"] = stats()
    return out


# ---- render (device buffers: raw pointers, e.g. torch.Tensor.data_ptr()) -----------------------------

def raytrace_device(view, lights7, indirect, mode, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb=None, d_index=None,
                    d_dist=None, d_pos=None):
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32)
    if d_dist is None and d_pos is None:
        _check(load().mirt_raytrace_device(C.byref(view), larr, nl, _ptr(ind), int(mode), int(y0), int(y1),
                                           int(row_origin), d_xrgb, int(pitch_bytes), d_rgb, d_index))
    else:
        _check(load().mirt_raytrace_device_ex(C.byref(view), larr, nl, _ptr(ind), int(mode), int(y0), int(y1),
                                              int(row_origin), d_xrgb, int(pitch_bytes), d_rgb, d_index, d_dist, d_pos))


def rasterise_device(view, lights7, indirect, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb=None, d_zinv=None,
                     d_index=None):
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32)
    _check(load().mirt_rasterise_device(C.byref(view), larr, nl, _ptr(ind), int(y0), int(y1), int(row_origin),
                                        d_xrgb, int(pitch_bytes), d_rgb, d_zinv, d_index))


def prepared_raytrace_device(view, lights7, indirect, mode, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb=None,
                             d_index=None):
    """Marshals the arguments once and returns a zero-argument callable that enqueues the frame: keeps the
    per-frame host cost of a render loop at one foreign call."""
    lib = load()
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32).copy()
    args = (C.byref(view), larr, nl, _ptr(ind), int(mode), int(y0), int(y1), int(row_origin), d_xrgb, int(pitch_bytes),
            d_rgb, d_index)
    fn = lib.mirt_raytrace_device

    def launch(_keep=(view, larr, ind)):
        rc = fn(*args)
        if rc:
            _check(rc)
    return launch


def prepared_rasterise_device(view, lights7, indirect, y0, y1, row_origin, d_xrgb, pitch_bytes, d_rgb=None, d_zinv=None,
                              d_index=None):
    lib = load()
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32).copy()
    args = (C.byref(view), larr, nl, _ptr(ind), int(y0), int(y1), int(row_origin), d_xrgb, int(pitch_bytes), d_rgb,
            d_zinv, d_index)
    fn = lib.mirt_rasterise_device

    def launch(_keep=(view, larr, ind)):
        rc = fn(*args)
        if rc:
            _check(rc)
    return launch


def prepared_async(kind, view, lights7, indirect, mode, surface):
    """Zero-argument callable: one asynchronous frame into `surface`, a numpy array inside a registered surface
    (mirt_raytrace_async / mirt_rasterise_async); mirt.sync() completes it."""
    lib = load()
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32).copy()
    if kind == "rt":
        fn, args = lib.mirt_raytrace_async, (C.byref(view), larr, nl, _ptr(ind), int(mode), _ptr(surface), int(surface.strides[0]))
    else:
        fn, args = lib.mirt_rasterise_async, (C.byref(view), larr, nl, _ptr(ind), _ptr(surface), int(surface.strides[0]))

    def launch(_keep=(view, larr, ind, surface)):
        rc = fn(*args)
        if rc:
            _check(rc)
    return launch


# ---- several GPUs: band sharding with the gather inside the library ------------------------------------

COMM_ID_BYTES = 128


def band_of(rank, world, height):
    y0, y1 = C.c_int(), C.c_int()
    _check(load().mirt_band_of(int(rank), int(world), int(height), C.byref(y0), C.byref(y1)))
    return y0.value, y1.value


def band_plan(world, root, width, height, nviews):
    """[(root_offset, band_offset, bytes, peer)] of one gather (mirt_band_plan)."""
    n = load().mirt_band_plan(world, root, width, height, nviews, None, None, None, None, 0)
    if n < 0:
        _check(n)
    ro, bo, by, pe = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.int32)
    load().mirt_band_plan(world, root, width, height, nviews, _ptr(ro), _ptr(bo), _ptr(by), _ptr(pe), n)
    return [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(ro, bo, by, pe)]


def set_partition(strip_rows):
    """0: contiguous bands; > 0: interleaved strips of that many rows (mirt_set_partition)."""
    _check(load().mirt_set_partition(int(strip_rows)))


def partition_segments(rank, world, height, strip_rows):
    """[(y0, y1)] of a rank's rows (mirt_partition_segments)."""
    n = load().mirt_partition_segments(rank, world, height, strip_rows, None, None, 0)
    if n < 0:
        _check(n)
    a, b = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    load().mirt_partition_segments(rank, world, height, strip_rows, _ptr(a), _ptr(b), n)
    return [(int(x), int(y)) for x, y in zip(a[:n], b[:n])]


def partition_plan(world, root, width, height, nviews, strip_rows):
    """[(root_offset, band_offset, bytes, peer)] of one gather for either partition (mirt_partition_plan)."""
    n = load().mirt_partition_plan(world, root, width, height, nviews, strip_rows, None, None, None, None, 0)
    if n < 0:
        _check(n)
    ro, bo, by, pe = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.int32)
    load().mirt_partition_plan(world, root, width, height, nviews, strip_rows, _ptr(ro), _ptr(bo), _ptr(by), _ptr(pe), n)
    return [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(ro[:n], bo[:n], by[:n], pe[:n])]


PARTITION_WEIGHTED = -1


def set_cost_histogram(on):
    """Binned ray-traced frames leave the cost histogram of the whole frame (mirt_set_cost_histogram)."""
    _check(load().mirt_set_cost_histogram(1 if on else 0))


def cost_histogram():
    """(hist, shift) of the latest binned frame -- estimated (tile, triangle) pairs per coarse tile row of (1 << shift) tile rows --
    or (None, 0)."""
    h = np.zeros(256, np.uint32)
    rows, shift = C.c_int(), C.c_int()
    n = load().mirt_cost_histogram(_ptr(h), 256, C.byref(rows), C.byref(shift))
    if n < 0:
        _check(n)
    return (h[:n].copy(), shift.value) if n > 0 else (None, 0)


def weighted_bounds(hist, shift, width, height, world):
    """world + 1 row boundaries of bands of equal estimated cost (mirt_weighted_bounds; pure arithmetic, no device)."""
    h = np.ascontiguousarray(hist if hist is not None else np.zeros(0), np.uint32)
    b = np.zeros(world + 1, np.int32)
    _check(load().mirt_weighted_bounds(_ptr(h) if len(h) else None, len(h), int(shift), int(width), int(height), int(world), _ptr(b)))
    return [int(x) for x in b]


def partition_bounds(world, width, height):
    """The boundaries the next sharded call will use (mirt_partition_bounds)."""
    b = np.zeros(world + 1, np.int32)
    _check(load().mirt_partition_bounds(int(world), int(width), int(height), _ptr(b)))
    return [int(x) for x in b]


def bounds_plan(world, root, width, height, nviews, bounds):
    """[(root_offset, band_offset, bytes, peer)] of one gather for explicit band boundaries (mirt_bounds_plan)."""
    bd = np.ascontiguousarray(bounds, np.int32)
    n = load().mirt_bounds_plan(world, root, width, height, nviews, _ptr(bd), None, None, None, None, 0)
    if n < 0:
        _check(n)
    ro, bo, by, pe = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.int32)
    load().mirt_bounds_plan(world, root, width, height, nviews, _ptr(bd), _ptr(ro), _ptr(bo), _ptr(by), _ptr(pe), n)
    return [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(ro[:n], bo[:n], by[:n], pe[:n])]


def comm_create_id():
    buf = (C.c_char * COMM_ID_BYTES)()
    _check(load().mirt_comm_create_id(buf))
    return bytes(buf)


def comm_init(comm_id, rank, world):
    buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(comm_id)
    _check(load().mirt_comm_init(buf, int(rank), int(world)))


def comm_shutdown():
    _check(load().mirt_comm_shutdown())


def comm_selfcheck(nbytes):
    _check(load().mirt_comm_selfcheck(C.c_size_t(int(nbytes))))


def view_array(views):
    arr = (View * len(views))()
    for i, v in enumerate(views):
        C.memmove(C.byref(arr[i]), C.byref(v), C.sizeof(View))
    return arr


def prepared_sharded(kind, views, lights7, indirect, mode, root, d_frames, pitch_bytes):
    """Zero-argument callable: this rank's band of len(views) frames + the gather on `root` (mirt_*_sharded)."""
    lib = load()
    arr = view_array(views)
    larr, nl = make_lights(lights7)
    ind = np.asarray(indirect, np.float32).copy()
    if kind == "rt":
        fn, args = lib.mirt_raytrace_sharded, (arr, len(views), larr, nl, _ptr(ind), int(mode), int(root), d_frames, int(pitch_bytes))
    else:
        fn, args = lib.mirt_rasterise_sharded, (arr, len(views), larr, nl, _ptr(ind), int(root), d_frames, int(pitch_bytes))

    def launch(_keep=(arr, larr, ind)):
        rc = fn(*args)
        if rc:
            _check(rc)
    return launch


def prepared_cull_device(view, flags=3):
    """Zero-argument callable that runs the cull step for `view` on the device (see prepared_raytrace_device)."""
    lib = load()
    ref = C.byref(view)
    fn = lib.mirt_cull_device

    def launch(_keep=(view,)):
        rc = fn(ref, int(flags))
        if rc:
            _check(rc)
    return launch


DEFAULT_LIGHT = np.array([[0.0, -0.5, -0.7, 1.0, 1.0, 1.0, 14.0]], np.float32)
