"""Regenerates the fixtures under tests/golden/ that come from code of the REAL reference.

Run in the build container (needs /root/reference): `make -C oracle && python tests/golden/make_golden.py`.
  cornell_ref.npy   LoadTestModel() of raytracer/Source/TestModel.h, 30 x 15 float32 {v0 v1 v2 normal color},
                    produced by oracle/_ref/libref_model.so (the reference header compiled in place).
survey_appendix_c.json is not generated here: it is transcribed from SURVEY.md Appendix C (outputs recorded
from the unmodified reference renderers, which cannot be rebuilt in this image -- they need SDL 1.2).
"""
import ctypes as C
import os

import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
lib = C.CDLL(os.path.join(root, "oracle", "_ref", "libref_model.so"))
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
lib.ref_load_test_model.restype = C.c_int
lib.ref_load_test_model.argtypes = [f32p, C.c_void_p, C.c_void_p]
out = np.zeros((30, 15), np.float32)
assert lib.ref_load_test_model(out, None, None) == 30
np.save(os.path.join(here, "cornell_ref.npy"), out)
print("wrote cornell_ref.npy", out.shape)
