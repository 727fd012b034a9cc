"""The CPU restatement under AddressSanitizer + UBSan (sanitizers run on the CPU build only).  Renders small frames of
both paths, including the awkward inputs (degenerate triangles, triangles through the camera plane, supersampling and
soft shadows), in a subprocess with the sanitizer runtime preloaded; any report fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from mirt_oracle import Oracle, DEFAULT_LIGHT
o = Oracle(%r)
t = o.cornell()
rot = o.rot_from_yaw(0.3, 1.0)
o.raytrace(t, (0, 0, -2), rot, 30.0, 61, 47, DEFAULT_LIGHT, threads=2)
s = o.soup(3, 300, 0.4)
s[5, 3:6] = s[5, 0:3]; s[9, 0:9] = np.tile(s[9, 0:3], 3)
import ctypes; ctypes.CDLL(None).srand(1)
jit = o.jitter(DEFAULT_LIGHT[0, 0:3], 4)
o.raytrace(s, (0.1, 0, -0.3), rot, 25.0, 50, 40, DEFAULT_LIGHT, threads=2, samples=4, jitter=jit, aa=3)
rotr = o.rot_from_yaw(-0.4, 1.01)
c = o.cull(s, (0, 0, -0.2), rotr, 60.0, 80, 60, 3)
o.rasterise(s, None, (0, 0, -0.2), rotr, 60.0, 80, 60, DEFAULT_LIGHT)
o.rasterise(t, o.cull(t, (0, 0, -3), rotr, 64.0, 64, 64, 3), (0, 0, -3), rotr, 64.0, 64, 64, DEFAULT_LIGHT)
print("sanitizer run ok")
'''


def test_oracle_clean_under_asan_ubsan():
    lib = os.path.join(ROOT, "oracle", "libmirt_oracle_asan.so")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmirt_oracle_asan.so"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(lib):
        pytest.skip("could not build the sanitizer variant: " + r.stderr[-300:])
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", CODE % (os.path.join(ROOT, "oracle"), lib)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "sanitizer run ok" in p.stdout, (p.stdout + p.stderr)[-3000:]
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
