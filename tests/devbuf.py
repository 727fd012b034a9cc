"""Device buffers for the *_device entry points of the C-ABI, without torch: plain hipMalloc / hipMemcpy through ctypes.
Test plumbing only."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        h = C.CDLL("libamdhip64.so")
        h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        h.hipFree.argtypes = [C.c_void_p]
        h.hipDeviceSynchronize.argtypes = []
        _hip = h
    return _hip


class DeviceArray:
    """A device plane shaped like a numpy array of `dtype`; `fill` is the byte it starts with."""

    def __init__(self, shape, dtype, fill=0):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = C.c_void_p()
        assert hip().hipMalloc(C.byref(self.ptr), max(self.nbytes, 16)) == 0
        assert hip().hipMemset(self.ptr, fill, self.nbytes) == 0
        assert hip().hipDeviceSynchronize() == 0

    def at(self, byte_offset):
        return C.c_void_p(self.ptr.value + int(byte_offset))

    def read(self):
        import mirt
        mirt.sync()
        out = np.zeros(self.shape, self.dtype)
        assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 2) == 0
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()
