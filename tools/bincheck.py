"""tools/bincheck.py -- how tight is the camera-frame binning?  (run on the GPU box)
With no lights the binned kernel's test counter is 64 x (sum of tile list lengths) = 64 x camera entries; the
exact number of 8x8-pixel tiles whose pixel-centre rectangle [8i,8i+7]x[8j,8j+7] overlaps each projected triangle
is computed here in float64 (separating axes: the two box axes and the three edge normals)."""
import sys
sys.path.insert(0, "cpp-raytracer-rasterizer_amd")
import numpy as np
import mirt

mirt.init(0)
W, H, f = 1920, 1080, 540.0
rot = np.zeros(9, np.float32); rot[0] = rot[4] = rot[8] = 1
view = mirt.make_view((0, 0, -2), rot, f, W, H)
soup = mirt.scene_soup(1, 100000, 0.05)


def exact_bins(t):
    V = t[:9].reshape(3, 3).astype(np.float64) - np.array([0, 0, -2.0])
    x = f * V[:, 0] / V[:, 2] + W / 2; y = f * V[:, 1] / V[:, 2] + H / 2
    i0 = int(np.ceil((x.min() - 7) / 8)); i1 = int(np.floor(x.max() / 8)); j0 = int(np.ceil((y.min() - 7) / 8)); j1 = int(np.floor(y.max() / 8))
    i0, j0 = max(i0, 0), max(j0, 0); i1, j1 = min(i1, W // 8 - 1), min(j1, H // 8 - 1)
    if i1 < i0 or j1 < j0:
        return 0, 0
    ii, jj = np.meshgrid(np.arange(i0, i1 + 1), np.arange(j0, j1 + 1)); ok = np.ones(ii.shape, bool)
    for a in range(3):
        b = (a + 1) % 3; c = (a + 2) % 3
        nx, ny = -(y[b] - y[a]), x[b] - x[a]; s = np.sign(nx * (x[c] - x[a]) + ny * (y[c] - y[a])); nx *= s; ny *= s
        xs = np.where(nx > 0, 8 * ii + 7, 8 * ii); ys = np.where(ny > 0, 8 * jj + 7, 8 * jj); ok &= (nx * (xs - x[a]) + ny * (ys - y[a]) >= 0)
    return int(ok.sum()), int(ii.size)


for n in (1000, 20000):
    sub = soup[:n].copy()
    mirt.scene_upload(sub)
    r = mirt.raytrace(view, np.zeros((0, 7), np.float32), mode=mirt.RT_BINNED, want_rgb=False, want_index=False)
    ex = [exact_bins(t) for t in sub]
    print("n=%d: GPU camera entries %.0f (%.2f/tri)  exact overlap %d (%.2f/tri)  bbox %d (%.2f/tri)" % (
        n, r["stats"]["tests"] / 64.0, r["stats"]["tests"] / 64.0 / n, sum(e for e, _ in ex), sum(e for e, _ in ex) / n,
        sum(b for _, b in ex), sum(b for _, b in ex) / n))
