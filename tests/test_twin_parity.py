"""The product's device code compiled for the CPU (tests/host_twin, test infrastructure) against
the independent oracle: every per-sample radiance and every ray count must be identical.  This is
the GPU-less rehearsal of tests/test_gpu_parity.py and also covers the orchestration logic
(pass planning, queue hand-offs, film gather) that both back ends share."""
import numpy as np
import pytest

import twin
from conftest import CORNELL


def _compare(orc, cam, scene, spp, depth, paths_per_pass=0):
    p = orc.make_params(cam.film.width, cam.film.height, spp, depth, paths_per_pass=paths_per_pass)
    fo, so, sto = orc.OracleScene(scene).render(cam, p, n_threads=4, want_samples=True)
    ft, stw, stt = twin.TwinScene(scene).render(cam, p, want_samples=True)
    assert (stt.samples, stt.rays_extension, stt.rays_shadow, stt.rays_mis) == (sto.samples, sto.rays_extension, sto.rays_shadow, sto.rays_mis)
    bad = (so.view(np.uint32) != stw.view(np.uint32)).any(axis=-1)
    assert bad.sum() == 0
    io, it = fo["rgb"] / fo["weight"][..., None], ft["rgb"] / ft["weight"][..., None]
    rel = np.sqrt(((io - it) ** 2).sum() / (io ** 2).sum())
    assert rel < 1e-5
    return ft, stw


@pytest.mark.parametrize("res,spp,depth", [((48, 48), 8, 15), ((33, 21), 2, 3), ((16, 16), 1, 0)])
def test_twin_cornell(ptrs, orc, res, spp, depth):
    cam, scene = ptrs.import_scene(CORNELL, res)
    _compare(orc, cam, scene, spp, depth)


def test_twin_material_zoo(scenes, orc):
    cam, scene = scenes.material_zoo((60, 40))
    _compare(orc, cam, scene, 4, 15)


def test_twin_soup_and_reference_tree(scenes, orc):
    cam, scene = scenes.triangle_soup(4000, resolution=(32, 32))
    _compare(orc, cam, scene, 4, 6)
    # the oracle's (reference-layout) tree handed through PtrsSceneDesc::bvh_nodes: same hits; the leaves
    # are visited in the same order, but both child boxes are tested when the parent is fetched (no
    # re-test against the shrunken t_max when a postponed child is popped), so the counters are close
    # to the reference's, not equal
    o = orc.OracleScene(scene)
    rng = np.random.default_rng(2)
    n = 3000
    org = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays = np.concatenate([org, d, np.full((n, 1), np.inf, np.float32)], axis=1)
    ref, ost = o.trace_rays(rays)
    t_own, _ = twin.TwinScene(scene).trace_rays(rays)
    t_ref, tst = twin.TwinScene(scene, bvh=o.get_bvh()).trace_rays(rays)
    for got in (t_own, t_ref):
        assert np.array_equal(got["prim"], ref["prim"])
        hit = ref["prim"] >= 0
        for f in ("t", "b0", "b1", "b2"):
            assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32))
    assert ost.tris_tested <= tst.tris_tested <= 1.15 * ost.tris_tested
    assert 0.8 * ost.nodes_visited <= tst.nodes_visited <= 1.2 * ost.nodes_visited


def test_twin_pass_and_band_decomposition(ptrs, orc):
    cam, scene = ptrs.import_scene(CORNELL, (40, 36))
    film_full, s_full = _compare(orc, cam, scene, 4, 5)
    film_small, s_small = _compare(orc, cam, scene, 4, 5, paths_per_pass=700)  # many row/sample chunks
    assert np.array_equal(s_full.view(np.uint32), s_small.view(np.uint32))
    tw = twin.TwinScene(scene)
    film = np.zeros((36, 40), dtype=film_full.dtype)
    for (a, b) in [(0, 13), (13, 30), (30, 36)]:
        tw.render(cam, orc.make_params(40, 36, 4, 5, row_begin=a, row_end=b), film=film)
    assert np.array_equal(film["rgb"].view(np.uint32), film_full["rgb"].view(np.uint32))
    assert np.array_equal(film["weight"].view(np.uint32), film_full["weight"].view(np.uint32))


def test_twin_sobol(orc):
    rng = np.random.default_rng(4)
    p = orc.make_params(1024, 1024, 256, 15)
    px, py = rng.integers(-2, 1026, 5000), rng.integers(-2, 1026, 5000)
    sn, dm = rng.integers(0, 256, 5000), rng.integers(0, 140, 5000)
    a, ia = twin.sobol_samples(p, px, py, sn, dm)
    b, ib = orc.sobol_samples(p, px, py, sn, dm)
    assert np.array_equal(ia, ib) and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_twin_textured_env(scenes, orc):
    """Image textures (resampled non-power-of-two pyramid, trilinear lookups at the camera vertex),
    a normal-mapped material, a float roughness texture and the importance-sampled environment
    light: the FEAT_FULL instantiations of the device code."""
    cam, scene = scenes.textured_env((72, 48))
    _compare(orc, cam, scene, 4, 15)


def test_twin_classroom_small(scenes, orc):
    """Reduced-detail classroom stand-in (glass slabs + sphere, Disney dielectrics, 512x1024 env map)."""
    cam, scene = scenes.classroom((64, 36), detail=0.02)
    _compare(orc, cam, scene, 2, 15)
