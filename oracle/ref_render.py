"""ctypes front-end of oracle/_ref/libref_rt*.so and libref_raster.so: the reference's own text of the hot-path functions
(oracle/ref_rt.cpp, oracle/ref_raster.cpp, oracle/extract_ref.py).  TEST INFRASTRUCTURE ONLY -- used by
tests/test_oracle_ref_render.py to pin oracle/mirt_oracle.c and by tests/golden/make_golden.py to record fixtures.
Frames of the reference are square (row stride SCREEN_HEIGHT): 500x500, or 150x150 for the
-DREALTIME build."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_vp = C.c_void_p
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def available():
    return all(os.path.exists(os.path.join(_HERE, "_ref", n)) for n in ("libref_rt.so", "libref_rt_150.so", "libref_raster.so"))


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


class RefRayTracer:
    """size = 500 (default build) or 150 (-DREALTIME)."""

    def __init__(self, size=500):
        self.lib = lib = C.CDLL(os.path.join(_HERE, "_ref", "libref_rt.so" if size == 500 else "libref_rt_150.so"))
        w, h, f, cam = C.c_int(), C.c_int(), C.c_float(), np.zeros(3, np.float32)
        lib.ref_rt_size.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), f32p]
        lib.ref_rt_size(C.byref(w), C.byref(h), C.byref(f), cam)
        assert w.value == h.value == size
        self.W = self.H = size
        self.default_focal, self.default_cam = f.value, cam
        lib.ref_rt_set_scene.argtypes = [f32p, C.c_int]
        lib.ref_rt_load_test_model.argtypes = [f32p]
        lib.ref_rt_set_view.argtypes = [f32p, C.c_float, C.c_float, f32p]
        lib.ref_rt_srand.argtypes = [C.c_uint]
        lib.ref_rt_add_light.argtypes = [f32p, f32p, C.c_float]
        lib.ref_rt_random_positions.argtypes = [f32p, C.c_int]
        lib.ref_rt_set_indirect.argtypes = [f32p]
        lib.ref_rt_set_options.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        lib.ref_rt_closest.argtypes = [f32p, f32p, C.c_int, f32p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        lib.ref_rt_direct_light.argtypes = [f32p, C.c_float, C.c_int, f32p]
        lib.ref_rt_draw.argtypes = [_vp] * 5
        lib.ref_rt_blur.argtypes = [f32p, f32p, f32p]
        self.samples = 16                      # SOFT_SHADOWS_SAMPLES as the library holds it (AddLight draws that many positions)

    def load_test_model(self):
        t = np.zeros((64, 15), np.float32)
        n = self.lib.ref_rt_load_test_model(t)
        return t[:n].copy()

    def set_scene(self, tris):
        t = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        self.lib.ref_rt_set_scene(t, len(t))

    def set_view(self, pos, yaw, focal):
        rot = np.zeros(9, np.float32)
        self.lib.ref_rt_set_view(np.asarray(pos, np.float32), float(yaw), float(focal), rot)
        return rot

    def set_options(self, aa=1, soft=1, dof=0, focal_plane=1.3, threads=8):
        self.lib.ref_rt_set_options(int(aa), int(soft), int(dof), float(focal_plane), int(threads))
        if soft > 1:
            self.samples = soft

    def set_lights(self, lights7, seed=1):
        """AddLight for every row of lights7 after srand(seed); returns randomPositions[0 : nlights*SOFT_SHADOWS_SAMPLES]."""
        l = np.ascontiguousarray(lights7, np.float32).reshape(-1, 7)
        self.lib.ref_rt_clear_lights()
        self.lib.ref_rt_srand(int(seed))
        for r in l:
            self.lib.ref_rt_add_light(r[0:3].copy(), r[3:6].copy(), float(r[6]))
        out = np.zeros((len(l) * self.samples, 3), np.float32)
        if len(l):
            self.lib.ref_rt_random_positions(out, len(out))
        return out

    def set_indirect(self, c):
        self.lib.ref_rt_set_indirect(np.asarray(c, np.float32))

    def closest(self, start, direction, is_light=False, pos=(0, 0, 0), distance=np.finfo(np.float32).max, index=-1):
        p, d, i = np.asarray(pos, np.float32).copy(), C.c_float(distance), C.c_int(index)
        any_ = self.lib.ref_rt_closest(np.asarray(start, np.float32), np.asarray(direction, np.float32), 1 if is_light else 0, p, C.byref(d), C.byref(i))
        return bool(any_), p, np.float32(d.value), i.value

    def direct_light(self, pos, distance, index):
        out = np.zeros(3, np.float32)
        self.lib.ref_rt_direct_light(np.asarray(pos, np.float32), float(distance), int(index), out)
        return out

    def draw(self):
        H, W = self.H, self.W
        out = {"index": np.zeros((H, W), np.int32), "dist": np.zeros((H, W), np.float32), "pos": np.zeros((H, W, 3), np.float32),
               "rgb": np.zeros((H, W, 3), np.float32), "fd": np.zeros((H, W), np.float32)}
        self.lib.ref_rt_draw(_p(out["index"]), _p(out["dist"]), _p(out["pos"]), _p(out["rgb"]), _p(out["fd"]))
        return out

    def blur(self, rgb, fd):
        out = np.zeros((self.H, self.W, 3), np.float32)
        self.lib.ref_rt_blur(np.ascontiguousarray(rgb, np.float32).reshape(-1), np.ascontiguousarray(fd, np.float32).reshape(-1), out.reshape(-1))
        return out


class RefRasteriser:
    def __init__(self):
        self.lib = lib = C.CDLL(os.path.join(_HERE, "_ref", "libref_raster.so"))
        w, h = C.c_int(), C.c_int()
        lib.ref_ra_size(C.byref(w), C.byref(h))
        self.W, self.H = w.value, h.value
        lib.ref_ra_set_scene.argtypes = [f32p, C.c_int]
        lib.ref_ra_load_test_model.argtypes = [f32p]
        lib.ref_ra_load_stl.argtypes = [C.c_char_p, _vp, C.c_int]
        lib.ref_ra_set_lights.argtypes = [_vp, C.c_int]
        lib.ref_ra_update.argtypes = [f32p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, f32p, f32p, _vp]
        lib.ref_ra_draw.argtypes = [_vp] * 3
        lib.ref_ra_vertex_shader.argtypes = [f32p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), f32p]
        lib.ref_ra_blur.argtypes = [f32p, f32p, C.c_int, f32p]
        self.n = 0

    def load_test_model(self):
        t = np.zeros((64, 15), np.float32)
        self.n = self.lib.ref_ra_load_test_model(t)
        return t[:self.n].copy()

    def load_stl(self, rasteriser_dir):
        n = self.lib.ref_ra_load_stl(str(rasteriser_dir).encode(), None, 0)
        if n < 0:
            raise OSError("cannot run LoadSTLFile from %s" % rasteriser_dir)
        t = np.zeros((n, 15), np.float32)
        self.lib.ref_ra_load_stl(str(rasteriser_dir).encode(), _p(t), n)
        return t

    def set_scene(self, tris):
        t = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        self.n = len(t)
        self.lib.ref_ra_set_scene(t, len(t))

    def set_lights(self, lights7):
        l = np.ascontiguousarray(lights7, np.float32).reshape(-1, 7)
        self.lib.ref_ra_set_lights(_p(l) if len(l) else None, len(l))

    def update(self, pos, yaw, focal, rot11=1.01, backface=True, frustum=True, focal_plane=1.9, indirect=(0.2, 0.2, 0.2)):
        """Update()'s `if (isUpdated)` body: returns (cameraRot as 9 floats, isCulled flags)."""
        rot, culled = np.zeros(9, np.float32), np.zeros(self.n, np.uint8)
        self.lib.ref_ra_update(np.asarray(pos, np.float32), float(yaw), float(focal), float(rot11), int(backface), int(frustum),
                               float(focal_plane), np.asarray(indirect, np.float32), rot, _p(culled))
        return rot, culled

    def draw(self):
        H, W = self.H, self.W
        out = {"depth": np.zeros((H, W), np.float32), "rgb": np.zeros((H, W, 3), np.float32), "fd": np.zeros((H, W), np.float32)}
        self.lib.ref_ra_draw(_p(out["depth"]), _p(out["rgb"]), _p(out["fd"]))
        return out

    def vertex_shader(self, v):
        x, y, z, p = C.c_int(), C.c_int(), C.c_float(), np.zeros(3, np.float32)
        self.lib.ref_ra_vertex_shader(np.asarray(v, np.float32), C.byref(x), C.byref(y), C.byref(z), p)
        return x.value, y.value, np.float32(z.value), p

    def blur(self, rgb, fd, kernel):
        out = np.zeros((self.H, self.W, 3), np.float32)
        self.lib.ref_ra_blur(np.ascontiguousarray(rgb, np.float32).reshape(-1), np.ascontiguousarray(fd, np.float32).reshape(-1), int(kernel), out.reshape(-1))
        return out
