"""Recorded outputs of the reference's own text (tests/golden/ref_render.json, made by tests/golden/make_golden.py from
oracle/_ref) for the frames of tests/ref_cases.py -- the pin that travels to machines without /root/reference.

  * without a GPU: the CPU restatement (oracle/mirt_oracle.c) reproduces every recorded buffer hash;
  * -m gpu: the HIP path, through the C-ABI, reproduces them directly -- closest-hit index, distance, position and float
    colours of the ray tracer (every kernel the mode switch can pick), depthBuffer and pixelColours of the rasteriser --
    and the file's Appendix C hashes equal the ones SURVEY.md transcribed from the whole unmodified program.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from ref_cases import RT_CASES, RASTER_CASES, build_scene, lights_array

HERE = os.path.dirname(os.path.abspath(__file__))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def rec():
    with open(os.path.join(HERE, "golden", "ref_render.json")) as f:
        return json.load(f)


def _jitter(e):
    return np.array(e["jitter"], np.float64).astype(np.float32).reshape(-1, 3) if "jitter" in e else None


def test_recipe_reproduces_survey_appendix_c(rec, golden):
    """The committed recipe regenerates the hashes SURVEY.md Appendix C holds (there transcribed from a run of the whole
    program): index map, pixelColours with and without the light, the rasteriser's depthBuffer and pixelColours."""
    g, r = golden["raytracer"], rec["raytracer"]
    assert r["cornell500_default"]["survey_appendix_c_fnv"] == {"index": g["index_fnv"], "rgb": g["with_light"]["rgb_fnv"]}
    assert r["cornell500_nolight"]["survey_appendix_c_fnv"] == {"index": g["index_fnv"], "rgb": g["no_light"]["rgb_fnv"]}
    assert r["cornell500_default"]["hits"] == g["hits"]
    gr = golden["rasteriser"]
    assert rec["rasteriser"]["cornell_default"]["survey_appendix_c_fnv"] == {"depth": gr["depth_fnv"], "rgb": gr["rgb_fnv"]}
    assert rec["rasteriser"]["cornell_default"]["covered"] == gr["covered"]
    assert rec["rasteriser"]["cornell_default"]["culled"] == gr["culled"]


@pytest.mark.parametrize("name", sorted(RT_CASES))
def test_oracle_reproduces_recorded_raytracer_frames(oracle, rec, name):
    c, e = RT_CASES[name], rec["raytracer"][name]
    S = c["size"]
    tris = build_scene(oracle, c["scene"])
    assert sha(tris) == e["scene_sha256"]
    rot = oracle.rot_from_yaw(c["yaw"], 1.0)
    assert rot.tolist() == e["rot"]
    got = oracle.raytrace(tris, c["cam"], rot, c["focal"], S, S, lights_array(c["lights"]), threads=8, samples=c["soft"], jitter=_jitter(e), aa=c["aa"])
    got["fd"] = np.where(got["index"] >= 0, got["dist"] - np.float32(1.3), np.float32(0)).astype(np.float32)
    for k, h in e["sha256"].items():
        assert sha(got[k]) == h, "%s: %s differs from the reference's recorded output" % (name, k)
    for s in e["samples"]:
        assert int(got["index"][s["y"], s["x"]]) == s["index"] and np.float32(s["dist"]) == got["dist"][s["y"], s["x"]]


@pytest.mark.parametrize("name", sorted(RASTER_CASES))
def test_oracle_reproduces_recorded_rasteriser_frames(oracle, rec, name):
    c, e = RASTER_CASES[name], rec["rasteriser"][name]
    tris = build_scene(oracle, c["scene"])
    assert sha(tris) == e["scene_sha256"]
    rot = oracle.rot_from_yaw(c["yaw"], c["rot11"])
    assert rot.tolist() == e["rot"]
    culled = oracle.cull(tris, c["cam"], rot, c["focal"], 500, 500, c["flags"])
    assert sha(culled) == e["culled_sha256"]
    got = oracle.rasterise(tris, culled, c["cam"], rot, c["focal"], 500, 500, lights_array(c["lights"]), want=("rgb", "index", "fd"), focal_plane=c["focal_plane"])
    for k, h in e["sha256"].items():
        assert sha(got[k]) == h, "%s: %s differs from the reference's recorded output" % (name, k)
    assert int((got["depth"] > 0).sum()) == e["covered"]


# ---- the HIP path against the same records ------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def device():
    import mirt
    mirt.init(0)
    yield mirt
    mirt.shutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "brute", "binned"])
@pytest.mark.parametrize("name", sorted(RT_CASES))
def test_gpu_reproduces_recorded_raytracer_frames(oracle, rec, device, name, mode):
    mirt = device
    c, e = RT_CASES[name], rec["raytracer"][name]
    S = c["size"]
    tris = build_scene(oracle, c["scene"])
    lights = lights_array(c["lights"])
    mirt.scene_upload(tris)
    mirt.set_soft_shadows(c["soft"], _jitter(e))
    mirt.set_antialiasing(c["aa"])
    try:
        view = mirt.make_view(c["cam"], np.array(e["rot"], np.float32), c["focal"], S, S)
        got = mirt.raytrace(view, lights, mode={"auto": mirt.RT_AUTO, "brute": mirt.RT_BRUTE, "binned": mirt.RT_BINNED}[mode], want_intersection=True)
    finally:
        mirt.set_soft_shadows(1)
        mirt.set_antialiasing(1)
    for k in ("index", "dist", "pos", "rgb"):
        assert sha(got[k]) == e["sha256"][k], "%s (%s): %s differs from the reference's recorded output" % (name, mode, k)
    assert got["stats"]["shadow_rays"] == e["hits"] * len(lights) * c["soft"] if c["aa"] == 1 else True


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(RASTER_CASES))
def test_gpu_reproduces_recorded_rasteriser_frames(oracle, rec, device, name):
    mirt = device
    c, e = RASTER_CASES[name], rec["rasteriser"][name]
    tris = build_scene(oracle, c["scene"])
    view = mirt.make_view(c["cam"], np.array(e["rot"], np.float32), c["focal"], 500, 500)
    mirt.scene_upload(tris)
    mirt.cull_device(view, c["flags"])
    assert sha(mirt.scene_get_culled()) == e["culled_sha256"]
    got = mirt.rasterise(view, lights_array(c["lights"]))
    assert sha(got["depth"]) == e["sha256"]["depth"] and sha(got["rgb"]) == e["sha256"]["rgb"]
    assert int((got["depth"] > 0).sum()) == e["covered"]
