"""Host-side texture / environment-light construction (pathtracer-rs_amd/textures.py)."""
import importlib
import os

import numpy as np
import pytest

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "abandoned_tank_farm_04_1k.hdr")  # data fixture: the map BASELINE configs[3] names


@pytest.fixture(scope="module")
def tx():
    return importlib.import_module("pathtracer-rs_amd.textures")


def test_mipmap_pyramid_shapes_and_box_filter(tx, ptrs):
    img = np.random.default_rng(0).uniform(0, 1, (8, 16, 3)).astype(np.float32)
    lv = tx.build_mipmap(img)
    assert [l.shape for l in lv] == [(8, 16, 3), (4, 8, 3), (2, 4, 3), (1, 2, 3), (1, 1, 3)]
    a = img
    ref = (((a[0::2, 0::2] + a[0::2, 1::2]) + a[1::2, 0::2]) + a[1::2, 1::2]) * np.float32(0.25)
    assert np.array_equal(lv[1], ref)
    assert abs(float(lv[-1].mean()) - float(img.mean())) < 1e-5
    # non power of two -> Lanczos resample to (16, 32) first (texture.rs:286-352)
    lv2 = tx.build_mipmap(np.ones((12, 20, 1), np.float32))
    assert lv2[0].shape == (16, 32, 1) and len(lv2) == 6


def test_env_distribution_is_normalised(tx, ptrs, scenes):
    s = ptrs.RenderScene()
    tx.add_infinite_light(s, scenes.synthetic_env_map())
    d = s.lights[0]["dist"]
    assert (d["nu"], d["nv"]) == (128, 64)
    assert np.allclose(d["cdf"][:, -1], 1.0, atol=1e-5) and np.all(np.diff(d["cdf"], axis=1) >= 0)
    assert abs(d["marg_cdf"][-1] - 1.0) < 1e-5 and d["marg_func_int"] > 0


def test_rgbe_reader_on_reference_asset(tx):
    """data/abandoned_tank_farm_04_1k.hdr (`-Y 512 +X 1024`, RLE): decoded like image 0.23's
    HdrDecoder (light.rs:331-346): c * 2^(e-136)."""
    img = tx.read_rgbe(HDR)
    assert img.shape == (512, 1024, 3) and img.dtype == np.float32
    assert np.isfinite(img).all() and img.min() >= 0 and 1e3 < img.max() < 1e6
    assert 0.2 < img.mean() < 2.0


def test_rgbe_reader_flat_and_rle_agree(tx, scenes, tmp_path):
    """The same image written as flat scanlines and as new-style RLE decodes to the same floats, c * 2^(e-136)."""
    from test_host_cpp import _write_rgbe
    img = scenes.synthetic_env_map(16, 32)
    img[3, 5:20] = img[3, 5]  # a long run
    a, b = str(tmp_path / "flat.hdr"), str(tmp_path / "rle.hdr")
    _write_rgbe(a, img)
    _write_rgbe(b, img, rle=True)
    assert os.path.getsize(b) != os.path.getsize(a)
    fa, fb = tx.read_rgbe(a), tx.read_rgbe(b)
    assert fa.shape == (16, 32, 3) and np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
    assert (np.abs(fa - img) <= img.max(axis=-1, keepdims=True) / 128).all()  # 8-bit mantissas under the pixel's shared exponent
