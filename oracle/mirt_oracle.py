"""ctypes front-end of the CPU restatement (oracle/mirt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py, never from the product package.  See the header of mirt_oracle.c for what pins it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmirt_oracle.so")

SURVEY_FNV_BASIS = 1469598103934665603   # see mirt_oracle_fnv1a64 in mirt_oracle.c
FNV_STD_BASIS = 0xcbf29ce484222325

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_vp = C.c_void_p


def build(force=False):
    """Compile the restatement (and, when /root/reference is present, oracle/_ref)."""
    src = os.path.join(_HERE, "mirt_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class Oracle:
    def __init__(self, path=None):
        self.lib = lib = C.CDLL(path or build())
        lib.mirt_oracle_fnv1a64.restype = C.c_uint64
        lib.mirt_oracle_fnv1a64.argtypes = [_vp, C.c_size_t, C.c_uint64]
        lib.mirt_oracle_rot_from_yaw.argtypes = [C.c_float, C.c_float, f32p]
        lib.mirt_oracle_cornell.restype = C.c_int
        lib.mirt_oracle_cornell.argtypes = [f32p]
        lib.mirt_oracle_soup.argtypes = [C.c_uint32, C.c_int, C.c_float, f32p]
        lib.mirt_oracle_raytrace.restype = C.c_uint64
        lib.mirt_oracle_raytrace.argtypes = [f32p, C.c_int, f32p, f32p, C.c_float, C.c_int, C.c_int, _vp, C.c_int,
                                             f32p, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int]
        lib.mirt_oracle_raytrace_soft.restype = C.c_uint64
        lib.mirt_oracle_raytrace_soft.argtypes = [f32p, C.c_int, f32p, f32p, C.c_float, C.c_int, C.c_int, _vp, C.c_int,
                                                  C.c_int, _vp, f32p, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int]
        lib.mirt_oracle_raytrace_ex.restype = C.c_uint64
        lib.mirt_oracle_raytrace_ex.argtypes = [f32p, C.c_int, f32p, f32p, C.c_float, C.c_int, C.c_int, _vp, C.c_int,
                                                C.c_int, _vp, C.c_int, f32p, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int]
        lib.mirt_oracle_jitter.argtypes = [f32p, C.c_int, f32p]
        lib.mirt_oracle_cull.argtypes = [f32p, C.c_int, f32p, f32p, C.c_float, C.c_int, C.c_int, C.c_int, _vp]
        lib.mirt_oracle_vertex_shader.argtypes = [f32p, f32p, f32p, C.c_float, C.c_int, C.c_int,
                                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float), f32p]
        lib.mirt_oracle_rasterise.argtypes = [f32p, _vp, C.c_int, f32p, f32p, C.c_float, C.c_int, C.c_int, _vp,
                                              C.c_int, f32p, _vp, _vp, _vp, _vp, C.c_int]
        lib.mirt_oracle_rasterise_ex.argtypes = [f32p, _vp, C.c_int, f32p, f32p, C.c_float, C.c_int, C.c_int, _vp,
                                                 C.c_int, f32p, C.c_float, _vp, _vp, _vp, _vp, _vp, C.c_int]
        lib.mirt_oracle_dof.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int]
        lib.mirt_oracle_load_stl.argtypes = [C.c_char_p, C.c_float, f32p, _vp, C.c_int]
        lib.mirt_oracle_closest_intersection.argtypes = [f32p, f32p, f32p, C.c_int, f32p, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        lib.mirt_oracle_direct_light.argtypes = [f32p, C.c_float, C.c_int, f32p, C.c_int, _vp, C.c_int, C.c_int, _vp, f32p]
        lib.mirt_oracle_dof_float.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, f32p, _vp]
        lib.mirt_oracle_load_stl.restype = C.c_int
        for name in ("mat3_inverse", "mat3_mul_vec", "vec_mul_mat3", "normalize", "cross"):
            getattr(lib, "mirt_oracle_" + name).argtypes = [f32p] * (2 if name in ("mat3_inverse", "normalize") else 3)
        lib.mirt_oracle_dot.restype = C.c_float
        lib.mirt_oracle_dot.argtypes = [f32p, f32p]
        lib.mirt_oracle_distance.restype = C.c_float
        lib.mirt_oracle_distance.argtypes = [f32p, f32p]

    # -- helpers -------------------------------------------------------------------------------
    def fnv(self, arr, basis=SURVEY_FNV_BASIS):
        """FNV-1a-64 of the raw bytes; default basis = the one SURVEY Appendix C's hashes were made with."""
        a = np.ascontiguousarray(arr)
        return int(self.lib.mirt_oracle_fnv1a64(_ptr(a), a.nbytes, basis))

    def rot_from_yaw(self, yaw, m11):
        r = np.zeros(9, np.float32)
        self.lib.mirt_oracle_rot_from_yaw(float(yaw), float(m11), r)
        return r

    def cornell(self):
        t = np.zeros((30, 15), np.float32)
        n = self.lib.mirt_oracle_cornell(t)
        assert n == 30
        return t

    def load_stl(self, path, scale=0.05, colour=(0.5, 0.5, 0.5)):
        """LoadSTL::LoadSTLFile: the triangles of an ASCII STL, scaled by -scale, grey, normals recomputed."""
        col = np.asarray(colour, np.float32)
        n = self.lib.mirt_oracle_load_stl(str(path).encode(), float(scale), col, None, 0)
        if n < 0:
            raise ValueError("cannot load %s (code %d)" % (path, n))
        t = np.zeros((n, 15), np.float32)
        self.lib.mirt_oracle_load_stl(str(path).encode(), float(scale), col, _ptr(t), n)
        return t

    def soup(self, seed, n, s):
        t = np.zeros((n, 15), np.float32)
        self.lib.mirt_oracle_soup(int(seed), int(n), float(s), t)
        return t

    # -- render paths --------------------------------------------------------------------------
    def jitter(self, light_pos, samples):
        """AddLight's soft-shadow positions for one light, drawn from the C library's rand() stream."""
        out = np.zeros((samples, 3), np.float32)
        self.lib.mirt_oracle_jitter(np.asarray(light_pos, np.float32), int(samples), out)
        return out

    def raytrace(self, tris, cam_pos, rot9, focal, W, H, lights, indirect=(0.2, 0.2, 0.2), y0=0, y1=None,
                 threads=0, want=("rgb", "index", "dist", "pos", "xrgb"), samples=1, jitter=None, aa=1):
        """Returns dict(rgb, index, dist, pos, xrgb, nshadow); full-frame planes, band rows filled.
        samples > 1: soft shadows with `jitter` = (nlights*samples, 3) positions; aa > 1: aa x aa supersampling."""
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        y1 = H if y1 is None else y1
        out = {
            "rgb": np.zeros((H, W, 3), np.float32) if "rgb" in want else None,
            "index": np.full((H, W), -1, np.int32) if "index" in want else None,
            "dist": np.zeros((H, W), np.float32) if "dist" in want else None,
            "pos": np.zeros((H, W, 3), np.float32) if "pos" in want else None,
            "xrgb": np.zeros((H, W), np.uint32) if "xrgb" in want else None,
        }
        jit = None if jitter is None else np.ascontiguousarray(jitter, np.float32)
        ns = self.lib.mirt_oracle_raytrace_ex(
            tris, len(tris), np.asarray(cam_pos, np.float32), np.ascontiguousarray(rot9, np.float32), float(focal),
            W, H, _ptr(lights) if len(lights) else None, len(lights), int(samples), _ptr(jit), int(aa),
            np.asarray(indirect, np.float32), y0, y1,
            threads, _ptr(out["rgb"]), _ptr(out["index"]), _ptr(out["dist"]), _ptr(out["pos"]), _ptr(out["xrgb"]), W)
        out["nshadow"] = int(ns)
        return out

    def closest_intersection(self, tris, start, direction, pos=(0, 0, 0), distance=np.finfo(np.float32).max, index=-1):
        """One ClosestIntersection call on an in/out Intersection record: (any, position, distance, index)."""
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        p, d, i = np.asarray(pos, np.float32).copy(), C.c_float(distance), C.c_int(index)
        any_ = self.lib.mirt_oracle_closest_intersection(np.asarray(start, np.float32), np.asarray(direction, np.float32), tris, len(tris),
                                                         p, C.byref(d), C.byref(i))
        return bool(any_), p, np.float32(d.value), i.value

    def direct_light(self, tris, pos, distance, index, lights, samples=1, jitter=None):
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        jit = None if jitter is None else np.ascontiguousarray(jitter, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.mirt_oracle_direct_light(np.asarray(pos, np.float32), float(distance), int(index), tris, len(tris),
                                          _ptr(lights) if len(lights) else None, len(lights), int(samples), _ptr(jit), out)
        return out

    def dof_float(self, rgb, fd, K):
        """CalculateDOF's blur as floats (what the reference hands to PutPixelSDL), interior pixels."""
        rgb = np.ascontiguousarray(rgb, np.float32)
        fd = np.ascontiguousarray(fd, np.float32)
        H, W = fd.shape
        out = np.zeros((H, W, 3), np.float32)
        scratch = np.zeros((H, W), np.uint32)
        self.lib.mirt_oracle_dof_float(rgb.reshape(-1), fd.reshape(-1), W, H, int(K), out.reshape(-1), _ptr(scratch))
        return out

    def cull(self, tris, cam_pos, rot9, focal, W, H, flags=3):
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        c = np.zeros(len(tris), np.uint8)
        self.lib.mirt_oracle_cull(tris, len(tris), np.asarray(cam_pos, np.float32),
                                  np.ascontiguousarray(rot9, np.float32), float(focal), W, H, flags, _ptr(c))
        return c

    def vertex_shader(self, v, cam_pos, rot9, focal, W, H):
        x, y, z = C.c_int(), C.c_int(), C.c_float()
        p = np.zeros(3, np.float32)
        self.lib.mirt_oracle_vertex_shader(np.asarray(v, np.float32), np.asarray(cam_pos, np.float32),
                                           np.ascontiguousarray(rot9, np.float32), float(focal), W, H,
                                           C.byref(x), C.byref(y), C.byref(z), p)
        return x.value, y.value, z.value, p

    def dof(self, rgb, fd, K, clear_border=False, xrgb=None, y0=0, y1=None):
        """CalculateDOF's blur + PutPixelSDL over rows [y0,y1); returns the surface words."""
        rgb = np.ascontiguousarray(rgb, np.float32)
        fd = np.ascontiguousarray(fd, np.float32)
        H, W = fd.shape
        out = np.zeros((H, W), np.uint32) if xrgb is None else xrgb
        self.lib.mirt_oracle_dof(rgb.reshape(-1), fd.reshape(-1), W, H, int(K), y0, H if y1 is None else y1,
                                 1 if clear_border else 0, _ptr(out), W)
        return out

    def rasterise(self, tris, culled, cam_pos, rot9, focal, W, H, lights, indirect=(0.2, 0.2, 0.2),
                  want=("rgb", "index", "xrgb"), focal_plane=0.0):
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 15)
        lights = np.ascontiguousarray(lights, np.float32).reshape(-1, 7)
        culled = None if culled is None else np.ascontiguousarray(culled, np.uint8)
        out = {
            "depth": np.zeros((H, W), np.float32),
            "rgb": np.zeros((H, W, 3), np.float32) if "rgb" in want else None,
            "index": np.full((H, W), -1, np.int32) if "index" in want else None,
            "xrgb": np.zeros((H, W), np.uint32) if "xrgb" in want else None,
            "fd": np.zeros((H, W), np.float32) if "fd" in want else None,
        }
        self.lib.mirt_oracle_rasterise_ex(
            tris, _ptr(culled), len(tris), np.asarray(cam_pos, np.float32), np.ascontiguousarray(rot9, np.float32),
            float(focal), W, H, _ptr(lights) if len(lights) else None, len(lights), np.asarray(indirect, np.float32),
            float(focal_plane), _ptr(out["depth"]), _ptr(out["rgb"]), _ptr(out["index"]), _ptr(out["fd"]), _ptr(out["xrgb"]), W)
        return out


# The reference's default light: AddLight(vec3(0,-0.5f,-0.7f), vec3(1,1,1), 14) (raytracer.cpp:116, rasteriser.cpp:104)
DEFAULT_LIGHT = np.array([[0.0, -0.5, -0.7, 1.0, 1.0, 1.0, 14.0]], np.float32)
