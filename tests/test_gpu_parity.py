"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
inputs.  Bars: bit-exact for integer work (Sobol indices, hit primitive ids, ray counts) and --
because host and device share one arithmetic definition -- also for every per-sample radiance;
the film (a sum whose order the reference itself does not fix, film.rs:213-228) is held to
1e-5 relative per-pixel L2 (north_star allows 1e-4)."""
import os

import numpy as np
import pytest

from conftest import CORNELL

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum() / max((b.astype(np.float64) ** 2).sum(), 1e-30)))


def test_sobol_device_matches_oracle(ptrs, orc):
    rng = np.random.default_rng(7)
    for (w, h, spp) in [(256, 256, 16), (1024, 1024, 256), (3840, 2160, 512), (5, 3, 1)]:
        p = orc.make_params(w, h, spp, 4)
        n = 20000
        px = rng.integers(-2, w + 2, n)
        py = rng.integers(-2, h + 2, n)
        sn = rng.integers(0, orc.round_up_pow2(spp), n)
        dims = rng.integers(0, 200, n)
        dims[:100] = 0
        dims[100:200] = 1
        got, gidx = ptrs.sobol_samples(p, px, py, sn, dims)
        ref, ridx = orc.sobol_samples(p, px, py, sn, dims)
        assert np.array_equal(gidx, ridx)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _camera_rays(cam, n, rng, w, h):
    # rays from the eye through random film points plus random interior rays
    o = np.tile(cam.trans, (n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 2, 2] = -8.0 * np.abs(d[: n // 2, 2]) - 4.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[n // 2:] = rng.uniform([-0.9, 0.1, -0.9], [0.9, 1.9, 0.9], size=(n - n // 2, 3)).astype(np.float32)
    t = np.full((n, 1), np.inf, dtype=np.float32)
    return np.concatenate([o, d.astype(np.float32), t], axis=1)


def test_trace_rays_matches_oracle(ptrs, orc):
    cam, scene = ptrs.import_scene(CORNELL, (64, 64))
    rng = np.random.default_rng(3)
    rays = _camera_rays(cam, 50000, rng, 64, 64)
    o = orc.OracleScene(scene)
    ref, _ = o.trace_rays(rays)
    got, st = ptrs.trace_rays(scene, rays)
    assert np.array_equal(got["prim"], ref["prim"])
    hit = ref["prim"] >= 0
    assert hit.mean() > 0.5
    for f in ("t", "b0", "b1", "b2"):
        assert np.array_equal(got[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), f
    # any-hit with finite segments
    rays2 = rays.copy()
    rays2[:, 6] = rng.uniform(0.1, 3.0, rays.shape[0]).astype(np.float32)
    ref2, _ = o.trace_rays(rays2, any_hit=True)
    got2, _ = ptrs.trace_rays(scene, rays2, any_hit=True)
    assert np.array_equal(got2["prim"], ref2["prim"])
    # the reference's own tree (handed over through PtrsSceneDesc::bvh_nodes) gives the same hits;
    # counters are close (both child boxes are tested at the parent), not equal
    nodes, prims = o.get_bvh()
    got3, st3 = ptrs.trace_rays(scene, rays, bvh=(nodes, prims))
    _, ost = o.trace_rays(rays)
    assert np.array_equal(got3["prim"], ref["prim"])
    assert ost.tris_tested <= st3.tris_tested <= 1.15 * ost.tris_tested
    assert 0.8 * ost.nodes_visited <= st3.nodes_visited <= 1.2 * ost.nodes_visited


@pytest.mark.parametrize("res,spp,depth", [((64, 64), 8, 15), ((256, 256), 16, 4), ((37, 23), 3, 2)])
def test_cornell_render_matches_oracle(ptrs, orc, res, spp, depth):
    cam, scene = ptrs.import_scene(CORNELL, res)
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(orc.round_up_pow2(spp), cam.film.get_sample_bounds()), depth)
    samples = integ.render(cam, scene, want_samples=True, flags=ptrs.abi.FLAG_COUNTERS)
    st = integ.last_stats
    o = orc.OracleScene(scene)
    film_ref, ref, ost = o.render(cam, orc.make_params(res[0], res[1], spp, depth), n_threads=8, want_samples=True)
    assert (st.samples, st.rays_extension, st.rays_shadow, st.rays_mis) == (ost.samples, ost.rays_extension, ost.rays_shadow, ost.rays_mis)
    bad = (samples.view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
    assert bad.sum() == 0, "%d of %d samples differ" % (bad.sum(), bad.size)
    img, img_ref = cam.film.to_rgb(), film_ref["rgb"] / film_ref["weight"][..., None]
    assert rel_l2(img, img_ref) < 1e-5
    assert np.allclose(cam.film.pixels["weight"], film_ref["weight"], rtol=1e-5)


def test_band_and_pass_decomposition_is_exact(ptrs, orc):
    """Multi-GPU row bands and the pass size must not change any value: samples bit-identical,
    film rows identical to the single-call render (each row's sum is formed in the same order)."""
    cam, scene = ptrs.import_scene(CORNELL, (96, 80))
    sb = cam.film.get_sample_bounds()
    full = ptrs.PathIntegrator(ptrs.SamplerBuilder(8, sb), 6)
    s_full = full.render(cam, scene, want_samples=True)
    film_full = cam.film.pixels.copy()
    cam.film.clear()
    small = ptrs.PathIntegrator(ptrs.SamplerBuilder(8, sb), 6, paths_per_pass=3000)
    s_small = small.render(cam, scene, want_samples=True)
    assert np.array_equal(s_full.view(np.uint32), s_small.view(np.uint32))
    assert rel_l2(cam.film.pixels["rgb"], film_full["rgb"]) < 1e-6
    cam.film.clear()
    for (a, b) in [(0, 27), (27, 64), (64, 80)]:
        full.render(cam, scene, row_begin=a, row_end=b)
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), film_full["rgb"].view(np.uint32))
    assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), film_full["weight"].view(np.uint32))


def _gpu_vs_oracle(ptrs, orc, cam, scene, spp, depth, **opts):
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(spp, cam.film.get_sample_bounds()), depth)
    with ptrs.options(**opts):
        samples = integ.render(cam, scene, want_samples=True)
    st = integ.last_stats
    if "lanes" in opts:
        assert st.lanes == opts["lanes"]
    if "tail_at" in opts:
        assert st.tail_launches == st.passes and st.tail_round == opts["tail_at"], "the scene has no fused-tail instantiation"
    film_ref, ref, ost = orc.OracleScene(scene).render(cam, orc.make_params(cam.film.width, cam.film.height, spp, depth), n_threads=8, want_samples=True)
    assert (st.samples, st.rays_extension, st.rays_shadow, st.rays_mis) == (ost.samples, ost.rays_extension, ost.rays_shadow, ost.rays_mis)
    bad = (samples.view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
    assert bad.sum() == 0, "%d of %d samples differ" % (bad.sum(), bad.size)
    assert rel_l2(cam.film.to_rgb(), film_ref["rgb"] / film_ref["weight"][..., None]) < 1e-5


def test_material_zoo_matches_oracle(ptrs, orc, scenes):
    """Mirror, glass (incl. the null-BSDF skip, Q7/Q17), metal, Disney, substrate, checker textures,
    point + directional + area lights: every per-sample radiance bit-identical to the oracle."""
    cam, scene = scenes.material_zoo((120, 80))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15)


def test_shipped_four_lane_schedule_matches_oracle(ptrs, orc, scenes):
    """Jobs under 4 M paths run on ONE pipeline lane with 16 384 segments; a full frame runs on FOUR lanes with 2 048 segments per pass
    (HipBackend::lanes).  Here the four-lane schedule itself is held against the oracle, sample by sample: the LDS-form scene
    (Cornell) and a quad-form scene (colonnade, tree in HBM), and the fused tail kernel of thin rounds with it."""
    cam, scene = ptrs.import_scene(CORNELL, (64, 64))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15, lanes=4)
    cam, scene = scenes.colonnade((160, 90))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 15, lanes=4)
    cam, scene = scenes.material_zoo((120, 80))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15, lanes=4)
    # how k_generate deals the chunks of a pass to the segments (round-robin, the default / by image region)
    cam, scene = ptrs.import_scene(CORNELL, (61, 47))  # (pixels % 64 != 0: the chunk matrix's rows drift)
    _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15, lanes=4, deal=1)
    cam, scene = scenes.colonnade((160, 90))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 15, lanes=4, deal=1)


def test_fused_tail_matches_oracle(ptrs, orc, scenes):
    """k_tail: from the hand-over round on every wave takes its queue segment through extend -> shade -> connect of ALL remaining
    rounds in one launch.  Held against the oracle sample by sample for every form it is instantiated for: the LDS form (Cornell),
    quad nodes out of global memory for a Matte scene (Cornell with node_form = 2) and for Disney + image textures (colonnade);
    from round 0 (the whole pass in one launch), from a middle round, on one lane and on four; and placed by the scene's
    learned survival profile (second render of a scene)."""
    def run(cam, scene, spp, depth, **opts):
        _gpu_vs_oracle(ptrs, orc, cam, scene, spp, depth, **opts)
    for opts in ({"tail_at": 0}, {"tail_at": 2, "lanes": 4}, {"tail_at": 7}, {"tail_at": 1, "node_form": 2}):
        cam, scene = ptrs.import_scene(CORNELL, (64, 64))  # (a fresh scene: node_form is read when the device scene is created, inside the render)
        run(cam, scene, 8, 15, **opts)
    # the learned placement: the first render of a scene runs round by round and leaves the profile, the second hands over where the
    # profile says a segment holds at most tail_paths paths -- here, with 16 384 segments for 37 k paths, at round 0
    cam, scene = ptrs.import_scene(CORNELL, (64, 64))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(8, cam.film.get_sample_bounds()), 15)
    integ.render(cam, scene)
    assert integ.last_stats.tail_launches == 0 and integ.last_stats.tail_round == 0xffffffff
    cam.film.clear()
    s2 = integ.render(cam, scene, want_samples=True)
    st = integ.last_stats
    assert st.tail_launches == st.passes and st.tail_round == 0
    _, ref, _ = orc.OracleScene(scene).render(cam, orc.make_params(64, 64, 8, 15), n_threads=8, want_samples=True)
    assert np.array_equal(s2.view(np.uint32), ref.view(np.uint32))
    cam, scene = scenes.colonnade((160, 90))
    run(cam, scene, 4, 15, tail_at=0)
    run(cam, scene, 4, 15, tail_at=3, lanes=4)


def test_triangle_soup_matches_oracle(ptrs, orc, scenes):
    """A 20k-triangle BVH (depth > 16: exercises the deeper LDS-stack instantiation)."""
    cam, scene = scenes.triangle_soup(20000, resolution=(64, 64))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 8)


def test_stack_spill_path_matches_oracle(ptrs, orc, scenes):
    """Deep trees spill stack entries beyond the 8-entry LDS column to global per-thread columns (the default for
    quad-form scenes, with the tree's top records cached in LDS); the option stack_lds = 16 is the layout without the
    cache.  Results must not depend on it."""
    cam, scene = scenes.triangle_soup(20000, resolution=(64, 64))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 8)
    with ptrs.options(stack_lds=16):
        cam, scene = scenes.triangle_soup(20000, resolution=(64, 64))
        _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 8)
    rng = np.random.default_rng(5)
    o = rng.uniform(-4, 4, (20000, 3)); d = rng.normal(size=(20000, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.full((20000, 1), np.inf)], axis=1).astype(np.float32)
    hg, _ = ptrs.trace_rays(scene, rays)
    ho, _ = orc.OracleScene(scene).trace_rays(rays)
    assert np.array_equal(hg["prim"], ho["prim"]) and np.array_equal(hg["t"].view(np.uint32), ho["t"].view(np.uint32))


def _lattice_scene(ptrs, n):
    """n^3 axis-aligned unit squares on integer planes (two triangles each): every box face of the tree lies in a coordinate plane."""
    s = ptrs.RenderScene()
    m = s.add_material(ptrs.abi.MAT_MATTE, [s.const_rgb([0.5, 0.5, 0.5])])
    pos, idx = [], []
    for i in range(n):
        for j in range(n):
            for k in range(n):
                ax = (i + j + k) % 3
                o = np.array([i, j, k], np.float32) * 2.0
                u, v = np.eye(3, dtype=np.float32)[(ax + 1) % 3], np.eye(3, dtype=np.float32)[(ax + 2) % 3]
                b = len(pos)
                pos += [o, o + u, o + u + v, o + v]
                idx += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    s.add_mesh(np.array(pos, np.float32), np.array(idx, np.uint32), m)
    return s


def test_rays_inside_box_planes(ptrs, orc):
    """Rays with zero direction components that start exactly in the planes of the tree's boxes: (plane - o) * (1 / 0) is
    0 * inf = NaN in the slab test.  The reference's compare-and-assign form keeps such a NaN, the kernels' v_max / v_min form
    needs its explicit `ordered` test to fail the same boxes (pt_bvh.h slab_entry6).  Pair form (3^3 squares fit LDS) and quad form."""
    dirs = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)], np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    for n, form in ((3, 0), (3, 2), (6, 0)):
        with ptrs.options(node_form=form):
            scene = _lattice_scene(ptrs, n)
            g = np.arange(-1, 2 * n + 1, 0.5, dtype=np.float32)
            rng = np.random.default_rng(11 + n)
            o = rng.choice(g, (6000, 3))
            d = dirs[rng.integers(0, len(dirs), 6000)]
            rays = np.concatenate([o, d, np.full((6000, 1), np.inf, np.float32)], axis=1).astype(np.float32)
            hg, _ = ptrs.trace_rays(scene, rays)
            ho, _ = orc.OracleScene(scene).trace_rays(rays)
            assert (ho["prim"] >= 0).sum() > 500
            assert np.array_equal(hg["prim"], ho["prim"]) and np.array_equal(hg["t"].view(np.uint32), ho["t"].view(np.uint32))
            occ, _ = ptrs.trace_rays(scene, rays, any_hit=True)
            oco, _ = orc.OracleScene(scene).trace_rays(rays, any_hit=True)
            assert np.array_equal(occ["prim"] >= 0, oco["prim"] >= 0)


def test_traversal_kernel_variants_give_the_same_samples(ptrs, orc, scenes):
    """The options refill / refill_connect = 0 select the fused k_extend / k_connect instead of the lane-refill kernels
    (+ k_epilogue / k_resolve), vote = 0 the while-while loop instead of phase voting, shade_lds = 0 global-memory tables in
    the shade kernels, fused_epilogue = 0 the separate k_epilogue / k_resolve passes; any combination and any idle-lane
    threshold must give the same samples."""
    cam, scene = scenes.triangle_soup(20000, resolution=(64, 64))
    for ext, con, vote, lds in ((0, 0, 1, 1), (1, 1, 1, 1), (0, 16, 0, 0), (64, 0, 1, 1), (16, 16, 0, 1), (16, 16, 1, 0), (48, 48, 1, 1), (16, 16, 2, 1), (16, 16, -1, 1)):
        with ptrs.options(refill=ext, refill_connect=con, vote=vote, shade_lds=lds, fused_epilogue=lds):
            _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 8)
    # and on an LDS-resident (pair-form) scene
    cam, scene = ptrs.import_scene(CORNELL, (48, 48))
    for ext, con, vote, lds in ((0, 0, 1, 1), (16, 16, 0, 0), (16, 16, 1, 1), (16, 0, 1, 0), (16, 0, 0, 1), (16, 16, 2, 1), (16, 16, -1, 1)):
        with ptrs.options(refill=ext, refill_connect=con, vote=vote, shade_lds=lds, fused_epilogue=lds):
            _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15)


def test_pipeline_lanes_do_not_change_the_film(ptrs):
    """Passes overlap on several pipeline lanes (streams with their own path state); the film kernels are chained
    in pass order, so the accumulators must be bit-identical to a single-lane render."""
    films = []
    for lanes in (1, 2, 4):
        with ptrs.options(lanes=lanes):
            cam, scene = ptrs.import_scene(CORNELL, (96, 80))
            integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(16, cam.film.get_sample_bounds()), 6, paths_per_pass=30000)  # 6+ passes
            integ.render(cam, scene)
            assert integ.last_stats.passes >= 6
            films.append(cam.film.pixels.copy())
    for f in films[1:]:
        assert np.array_equal(f["rgb"].view(np.uint32), films[0]["rgb"].view(np.uint32))
        assert np.array_equal(f["weight"].view(np.uint32), films[0]["weight"].view(np.uint32))


def test_launch_policy_options_do_not_change_the_film(ptrs):
    """How the queue kernels are launched -- segments per pass, persistent or maximal grids, the share of the resident capacity a launch
    takes, whole-round segment counts, lanes chosen by job size or given -- is scheduling only: the film is bit-identical for every
    combination, on the LDS form and on a quad-form scene with glass (spare rounds, deferred pass closing)."""
    def film(scene_fn, opts, spp, depth, ppp):
        with ptrs.options(**opts):
            cam, scene = scene_fn()
            integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(spp, cam.film.get_sample_bounds()), depth, paths_per_pass=ppp)
            integ.render(cam, scene)
            return cam.film.pixels.copy(), integ.last_stats
    scenes_mod = __import__("importlib").import_module("pathtracer-rs_amd.scenes")
    cases = [(lambda: ptrs.import_scene(CORNELL, (96, 80)), 16, 8, 30000), (lambda: scenes_mod.material_zoo((72, 48)), 8, 15, 20000)]
    combos = [{}, {"lanes": 1}, {"lanes": 4, "grid_pct": 10}, {"grid_mult": 1, "persist": 0}, {"grid_mult": 64, "whole_rounds": 1}, {"lanes": 3, "grid_pct": 100, "grid_mult": 3},
              {"lanes": 2, "fused_epilogue": 0, "grid_mult": 2}, {"lanes": 4, "node_form": 2}, {"lanes": 4, "node_form": 2, "grid_mult": 3},  # (node_form 2: quad nodes with per-axis planes out of global memory)
              {"deal": 0}, {"deal": 1}, {"deal": 1, "lanes": 4, "tail_at": 2}, {"deal": 1, "grid_mult": 3, "lanes": 2}, {"deal": 1, "node_form": 2, "lanes": 4},  # chunks dealt to segments round-robin / by image region
              {"tail": 0}, {"tail_at": 0}, {"tail_at": 1, "lanes": 4}, {"tail_at": 3, "grid_mult": 2}, {"tail_at": 5, "lanes": 1}, {"tail_at": 2, "node_form": 2, "lanes": 4}]  # the fused tail from round k on (Cornell has an instantiation, the zoo -- every material kind -- has none: the knob is inert there)
    for scene_fn, spp, depth, ppp in cases:
        ref, st0 = film(scene_fn, combos[0], spp, depth, ppp)
        assert st0.passes >= 2 and st0.queue_segments > 0 and st0.lanes >= 1
        for opts in combos[1:]:
            got, st = film(scene_fn, opts, spp, depth, ppp)
            assert np.array_equal(got["rgb"].view(np.uint32), ref["rgb"].view(np.uint32)), opts
            assert np.array_equal(got["weight"].view(np.uint32), ref["weight"].view(np.uint32)), opts
            assert (st.rays_extension, st.rays_shadow, st.rays_mis) == (st0.rays_extension, st0.rays_shadow, st0.rays_mis), opts
            if "lanes" in opts:
                assert st.lanes == opts["lanes"]


def test_full_size_properties(ptrs):
    """BASELINE configs[1] at full size (1024x1024, depth 15; 16 spp to stay within the test budget):
    size-independent properties -- filter-weight sums are the analytic constant in the interior,
    radiance finite and non-negative, and the result is reproducible bit for bit run to run."""
    cam, scene = ptrs.import_scene(CORNELL, (1024, 1024))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(16, cam.film.get_sample_bounds()), 15)
    integ.render(cam, scene)
    a = cam.film.pixels.copy()
    st = integ.last_stats
    assert st.samples == 1028 * 1028 * 16 and st.rays_extension >= st.samples
    assert st.lanes == 4 and st.passes == 4 and st.queue_segments % 8 == 0 and st.grid_pct == 100  # the default launch policy for a job of this size: four lanes of CUs x 8 wave-segments
    assert 0.0 < st.ms_enqueue <= st.ms_total  # the host is done enqueuing long before the device is through
    w = a["weight"][8:-8, 8:-8]
    assert np.isfinite(a["rgb"]).all() and (a["rgb"] >= 0).all()
    assert np.allclose(w, np.median(w), rtol=0.35) and w.min() > 0
    img = cam.film.to_rgb()
    assert 0.05 < img.mean() < 1.0
    assert st.tail_launches == 0  # the scene's first render: no survival profile yet, every round is its three launches
    cam.film.clear()
    integ.render(cam, scene)
    st2 = integ.last_stats
    assert st2.tail_launches == st2.passes and 1 <= st2.tail_round < 16 and st2.kernel_launches < st.kernel_launches  # ... the second hands its thin rounds to the fused tail
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), a["rgb"].view(np.uint32))
    assert (st2.rays_extension, st2.rays_shadow, st2.rays_mis) == (st.rays_extension, st.rays_shadow, st.rays_mis)
    cam.film.clear()
    with ptrs.options(lanes=1, deal=1, tail_at=3):  # one lane of 16 384 segments dealt by image region, the tail from round 3: the same film at full size
        integ.render(cam, scene)
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), a["rgb"].view(np.uint32))
    assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), a["weight"].view(np.uint32))


def test_textured_env_matches_oracle(ptrs, orc, scenes):
    """FEAT_FULL kernels on the GPU: MIP-mapped image textures, normal mapping, environment light
    (Distribution2D sampling, acos/atan2 stand-ins)."""
    cam, scene = scenes.textured_env((96, 64))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15)


def test_environment_light_presampling_variants(ptrs, orc, scenes):
    """The environment light's samples come from k_env_presample (default) or are evaluated inside the shade kernels
    (env_presample = 0), with the shade kernels' tables in LDS or in global memory: the same samples either way."""
    for pre, lds in ((0, 1), (1, 0), (0, 0)):
        with ptrs.options(env_presample=pre, shade_lds=lds):
            cam, scene = scenes.textured_env((96, 64))
            _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15)


def test_colonnade_matches_oracle(ptrs, orc, scenes):
    """Sponza-class stand-in (~262k triangles, BVH depth > 16, HBM-resident tree, Disney metal with an
    image texture, directional + point lights) at reduced resolution."""
    cam, scene = scenes.colonnade((160, 90))
    assert 240000 < scene.num_triangles() < 290000
    _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 15)


def test_classroom_matches_oracle(ptrs, orc, scenes):
    """Classroom-class stand-in (BASELINE configs[3]: ~600k triangles, solid glass panes + a glass sphere,
    Disney dielectrics with an image texture, lit only by a 512x1024 HDR environment map through the
    windows) at reduced resolution: FEAT_FULL kernels on an HBM-resident tree."""
    cam, scene = scenes.classroom((128, 72))
    assert 560000 < scene.num_triangles() < 650000
    _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 15)


def test_imported_gltf_matches_oracle(ptrs, orc, tmp_path):
    """The synthetic glTF asset (tests/gltf_fixture.py) imported by gltf.py: Disney + normal map + metallic-roughness
    textures, glass, mirror, alpha-masked card, constant and textured emissive quads, directional + point lights."""
    import gltf_fixture as gf
    cam, scene = ptrs.import_scene(gf.write_gltf(str(tmp_path), glb=True), (96, 64))
    _gpu_vs_oracle(ptrs, orc, cam, scene, 8, 15)


def test_film_matches_twin_bitwise(ptrs, orc):
    """The film kernel (LDS-tiled gather) forms every pixel's sums in the same order as the host twin's
    film_item (sample index, then sample-pixel x, then y), so the accumulators agree bit for bit --
    a stricter check of the film path than the 1e-5 comparison with the oracle's tile order."""
    import twin
    cam, scene = ptrs.import_scene(CORNELL, (70, 45))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(8, cam.film.get_sample_bounds()), 6, paths_per_pass=40000)
    integ.render(cam, scene)
    ft, _, _ = twin.TwinScene(scene).render(cam, orc.make_params(70, 45, 8, 6, paths_per_pass=40000))
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), ft["rgb"].view(np.uint32))
    assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), ft["weight"].view(np.uint32))


def _tiny_scene(ptrs, kind):
    """Degenerate inputs the reference's own code paths have to survive (and so must the GPU path)."""
    abi = ptrs.abi
    s = ptrs.RenderScene()
    white = s.add_material(abi.MAT_MATTE, [s.const_rgb([0.7, 0.7, 0.7])])
    tri = np.array([[-1, 0, -3], [1, 0, -3], [0, 1.5, -3]], np.float32)
    if kind == "one_triangle_emissive":      # whole scene = one leaf (root is a leaf): single-slot node
        s.add_mesh(tri, np.array([[0, 1, 2]], np.uint32), white, emission_rgb=[3.0, 2.0, 1.0])
    elif kind == "no_lights":                # uniform_sample_one_light returns before drawing samples
        s.add_mesh(tri, np.array([[0, 1, 2]], np.uint32), white)
        s.add_mesh(tri + np.array([0, 0, -1], np.float32), np.array([[0, 2, 1]], np.uint32), white)
    elif kind == "degenerate_and_duplicate":  # zero-area triangle, coincident duplicates (ties), point light
        pos = np.array([[-2, -1, -4], [2, -1, -4], [2, 2, -4], [-2, 2, -4], [0, 0, -3.5], [0, 0, -3.5], [0, 0, -3.5]], np.float32)
        idx = np.array([[0, 1, 2], [0, 2, 3], [0, 1, 2], [4, 5, 6], [0, 2, 3]], np.uint32)
        s.add_mesh(pos, idx, white)
        s.add_point_light([0.5, 0.5, -1.0], [8.0, 8.0, 8.0])
    elif kind == "null_bsdf_glass_stack":     # black glass = no BSDF: the path skips the surfaces with bounces -= 1 (Q7)
        glass = s.add_material(abi.MAT_GLASS, [s.const_rgb([0, 0, 0]), s.const_rgb([0, 0, 0]), s.const_f(1.5)])
        for k in range(5):
            s.add_mesh(tri * np.float32(2) + np.array([0, -1, 0.2 * k], np.float32), np.array([[0, 1, 2]], np.uint32), glass)
        s.add_mesh(tri * np.float32(4) + np.array([0, -2, -2], np.float32), np.array([[0, 1, 2]], np.uint32), white, emission_rgb=[2.0, 2.0, 2.0])
    cam = ptrs.look_at_camera([0.0, 0.5, 1.5], [0.0, 0.5, -3.0], [0, 1, 0], 50.0, (24, 16))
    return cam, s


@pytest.mark.parametrize("kind", ["one_triangle_emissive", "no_lights", "degenerate_and_duplicate", "null_bsdf_glass_stack"])
@pytest.mark.parametrize("spp,depth", [(1, 0), (4, 15)])
def test_edge_scenes_match_oracle(ptrs, orc, kind, spp, depth):
    cam, scene = _tiny_scene(ptrs, kind)
    _gpu_vs_oracle_film_tolerant(ptrs, orc, cam, scene, spp, depth)


def _gpu_vs_oracle_film_tolerant(ptrs, orc, cam, scene, spp, depth):
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(spp, cam.film.get_sample_bounds()), depth)
    samples = integ.render(cam, scene, want_samples=True)
    st = integ.last_stats
    film_ref, ref, ost = orc.OracleScene(scene).render(cam, orc.make_params(cam.film.width, cam.film.height, spp, depth), n_threads=4, want_samples=True)
    assert (st.samples, st.rays_extension, st.rays_shadow, st.rays_mis) == (ost.samples, ost.rays_extension, ost.rays_shadow, ost.rays_mis)
    assert np.array_equal(samples.view(np.uint32), ref.view(np.uint32))
    a, b = cam.film.pixels, film_ref
    assert np.allclose(a["weight"], b["weight"], rtol=1e-5) and np.allclose(a["rgb"], b["rgb"], rtol=1e-4, atol=1e-7)


def test_render_single_pixel_on_the_gpu(ptrs, orc):
    """PathIntegrator::render_single_pixel (integrator.rs:505-534) through ptrs_render_single_pixel."""
    cam, scene = ptrs.import_scene(CORNELL, (40, 40))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(8, cam.film.get_sample_bounds()), 6)
    p = orc.make_params(40, 40, 8, 6)
    O = orc.OracleScene(scene)
    for px, py in ((0, 0), (17, 23), (39, 39), (-2, 5), (41, 39), (5, -2), (5, 41), (-2, -2), (41, 41)):  # incl. the filter apron on all four sides
        got = integ.render_single_pixel(cam, (px, py), scene)
        want = O.render_single_pixel(cam, p, px, py)
        assert np.array_equal(np.asarray(got).view(np.uint32), want.view(np.uint32))
    assert integ.last_single_pixel_paths == 8  # only the pixel's spp paths were traced
    with pytest.raises(ptrs.PtrsError):  # outside the sample bounds
        integ.render_single_pixel(cam, (42, 0), scene)


def test_cfg5_band_with_33_bit_sobol_indices(ptrs, orc, scenes):
    """BASELINE configs[4] settings (3840x2160, 512 spp, depth 15: Sobol indices need 33 bits, m = 12) on a 2-row band
    of a reduced-detail colonnade: the whole pipeline with the index's high word in the path state, against the oracle."""
    cam, scene = scenes.colonnade((3840, 2160), detail=0.01, tex_size=64)
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(512, cam.film.get_sample_bounds()), 15)
    rb, re = 1000, 1002
    integ.render(cam, scene, row_begin=rb, row_end=re)  # (the per-sample export would be 51 GB at this size)
    st = integ.last_stats
    p = orc.make_params(3840, 2160, 512, 15, row_begin=rb, row_end=re)
    film_ref, _, ost = orc.OracleScene(scene).render(cam, p, n_threads=16)
    assert st.samples == ost.samples == 3844 * 6 * 512
    # identical ray counts over 11.8 M paths: every branch decision (hits, lobe picks, Russian roulette) agreed
    assert (st.rays_extension, st.rays_shadow, st.rays_mis) == (ost.rays_extension, ost.rays_shadow, ost.rays_mis)
    a, b = cam.film.pixels[rb:re], film_ref[rb:re]
    assert (b["weight"] > 0).all()
    # 12 800 binary32 additions per pixel in two different orders (the reference's own order depends on tile scheduling)
    assert np.allclose(a["weight"], b["weight"], rtol=1e-4) and rel_l2(a["rgb"] / a["weight"][..., None], b["rgb"] / b["weight"][..., None]) < 1e-4


def _band_vs_fixture(ptrs, cam, scene, spp, name):
    """Rows of a full-settings render against the oracle fixture tests/golden/bench_<name>_rows.npz (made by make_golden.py --bench):
    ray counts of the band identical, film rows within the tolerance of two summation orders."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_%s_rows.npz" % name))
    rb, re = int(z["row0"]), int(z["row1"])
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(spp, cam.film.get_sample_bounds()), 15)
    integ.render(cam, scene, row_begin=rb, row_end=re)
    st = integ.last_stats
    assert (st.rays_extension, st.rays_shadow, st.rays_mis, st.samples) == tuple(int(v) for v in z["rays"])
    a = cam.film.pixels[rb:re]
    got = np.concatenate([a["rgb"], a["weight"][..., None]], axis=-1)
    ref = z["film"]
    assert np.allclose(got[..., 3], ref[..., 3], rtol=1e-4)
    assert rel_l2(got[..., :3] / got[..., 3:], ref[..., :3] / ref[..., 3:]) < 1e-4


def test_cfg3_colonnade_band_at_full_settings(ptrs, scenes):
    """BASELINE configs[2] settings (1280x720, 64 spp, depth 15) on the 262 k-triangle colonnade stand-in, rows 358-360."""
    cam, scene = scenes.colonnade((1280, 720))
    _band_vs_fixture(ptrs, cam, scene, 64, "colonnade")


def test_cfg4_classroom_band_at_full_settings(ptrs, scenes):
    """BASELINE configs[3] settings (1920x1080, 128 spp, depth 15) on the 606 k-triangle classroom stand-in lit by
    data/abandoned_tank_farm_04_1k.hdr (the map the config names), rows 540-542."""
    cam, scene = scenes.classroom((1920, 1080))
    _band_vs_fixture(ptrs, cam, scene, 128, "classroom")


def test_progressive_render_publishes_the_film_pass_by_pass(ptrs):
    """ptrs_render_progressive: after each pass the touched rows are in the host film (at least that pass's samples), the
    callback sees every pass once and in order, and the final film is bit-identical to ptrs_render's."""
    cam, scene = ptrs.import_scene(CORNELL, (64, 48))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(16, cam.film.get_sample_bounds()), 6, paths_per_pass=20000)  # several passes
    integ.render(cam, scene)
    ref = cam.film.pixels.copy()
    n_ref = integ.last_stats.passes
    cam.film.clear()
    seen, sums = [], []

    def on_pass(done, total, y0, y1):
        seen.append((done, total, y0, y1))
        sums.append(float(cam.film.pixels["weight"].sum()))
    integ.render_progressive(cam, scene, on_pass)
    assert [d for d, _, _, _ in seen] == list(range(1, n_ref + 1)) and all(t == n_ref for _, t, _, _ in seen) and n_ref >= 4
    assert all(0 <= y0 < y1 <= 48 for _, _, y0, y1 in seen)
    # a pass's rows are copied out when its pipeline lane is next waited for: passes behind it may already have landed in them, so the
    # weights never shrink from callback to callback (and grow over the render), they need not grow at every single one
    assert all(b >= a for a, b in zip(sums, sums[1:])) and sums[0] > 0 and sums[-1] > sums[0]
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), ref["weight"].view(np.uint32))


def test_multi_device_render_is_bit_identical(ptrs):
    """ptrs_render_multi (one process, one host thread and one scene replica per device, bands gathered with device
    copies): 2 and 3 replicas -- all on the one GPU of this box -- with equal and with cost-weighted bands must reproduce
    the single-device film bit for bit, and the bands' ray counts (minus the re-traced halo rows) must add up."""
    cam, scene = ptrs.import_scene(CORNELL, (96, 80))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(16, cam.film.get_sample_bounds()), 8)
    integ.render(cam, scene)
    ref = cam.film.pixels.copy()
    for n, cost in ((2, None), (3, None), (3, np.linspace(1.0, 4.0, 80)), (4, np.r_[np.zeros(40), np.ones(40)])):
        cam.film.clear()
        bounds, stats = integ.render_multi(cam, scene, [0] * n, row_cost=cost)
        assert bounds[0] == 0 and bounds[-1] == 80 and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
        assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), ref["weight"].view(np.uint32))
        traced = sum((b1 - b0 + 4) for b0, b1 in zip(bounds, bounds[1:]) if b1 > b0)
        assert sum(st.samples for st in stats) == traced * 100 * 16  # each band traces its rows + 4 sample rows of apron / halo
    if cost is not None:
        assert bounds != [0, 20, 40, 60, 80]  # the weighted plan differs from equal bands
    # a cleared film the caller vouches for (PTRS_FLAG_FILM_ZERO: bands are cleared on the devices, not uploaded), and an
    # accumulating second render on top of the first (no flag: the host film's bands are uploaded)
    cam.film.clear()
    integ.render_multi(cam, scene, [0, 0, 0], film_is_zero=True)
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    integ.render_multi(cam, scene, [0, 0])
    cam2, _ = ptrs.import_scene(CORNELL, (96, 80))
    integ.render(cam2, scene); integ.render(cam2, scene)
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), cam2.film.pixels["rgb"].view(np.uint32))
    assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), cam2.film.pixels["weight"].view(np.uint32))


def test_multi_device_render_argument_checks_and_scene_options(ptrs):
    """ptrs_render_multi refuses the same scene handle twice (two host threads on one workspace would race); per-scene options
    (ptrs_scene_set_option) override the process-wide ones for that scene only and change no bit."""
    import ctypes as C
    from importlib import import_module
    integ_mod = import_module("pathtracer-rs_amd.integrator")
    cam, scene = ptrs.import_scene(CORNELL, (64, 48))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(4, cam.film.get_sample_bounds()), 5)
    integ.render(cam, scene)
    ref = cam.film.pixels.copy()
    L = ptrs.load_library()
    ds = integ_mod._DeviceScene(scene, 0)
    try:
        handles = (C.c_void_p * 2)(ds.handle, ds.handle)
        p, c = integ.params(cam), cam.to_abi()
        film = np.zeros_like(ref)
        rc = L.ptrs_render_multi(handles, 2, C.byref(c), C.byref(p), None, C.c_void_p(film.ctypes.data), None)
        assert rc != 0 and b"twice" in L.ptrs_last_error()
        with pytest.raises(ptrs.PtrsError):
            ds.set_option("lanes", 9)
        with pytest.raises(ptrs.PtrsError):
            ds.set_option("no_such_knob", 1)
    finally:
        ds.close()
    for opts in ({"lanes": 1, "grid_mult": 2}, {"vote": 0, "refill": 48, "grid_pct": 50}):
        cam.film.clear()
        integ.render_multi(cam, scene, [0, 0], scene_options=opts)
        assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    assert ptrs.get_option("lanes") == 0  # the process-wide value is untouched


def test_multi_device_host_staged_gather(ptrs):
    """The path ptrs_render_multi takes when two devices cannot reach each other -- bands staged through the host film -- forced
    with the scene option peer_copy = 0, which also applies between replicas on one device."""
    cam, scene = ptrs.import_scene(CORNELL, (64, 48))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(4, cam.film.get_sample_bounds()), 5)
    integ.render(cam, scene)
    ref = cam.film.pixels.copy()
    cam.film.clear()
    integ.render_multi(cam, scene, [0, 0, 0], scene_options={"peer_copy": 0})
    assert np.array_equal(cam.film.pixels["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    assert np.array_equal(cam.film.pixels["weight"].view(np.uint32), ref["weight"].view(np.uint32))


def test_sobol_dimension_overrun_is_an_error(ptrs):
    """The reference panics when a sample needs Sobol dimension >= 1024 (sobol.rs:177-183); a matte vertex draws 8, so a
    path deeper than ~127 bounces gets there.  Closed white box, Russian roulette off, max_depth 200: render must fail with
    PTRS_ERR_UNSUPPORTED instead of reading past the table (and succeed at a depth that stays inside it)."""
    s = ptrs.RenderScene()
    white = s.add_material(ptrs.abi.MAT_MATTE, [s.const_rgb([1.0, 1.0, 1.0])])
    from importlib import import_module
    sc = import_module("pathtracer-rs_amd.scene")
    pos, nrm, idx = sc.gen_cube()
    s.add_mesh((pos * np.float32(2.0)).astype(np.float32), idx, white, normal=(-nrm).astype(np.float32))
    s.add_point_light([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])
    cam = ptrs.look_at_camera([0.5, 0.3, 1.0], [0.0, 0.0, -1.0], [0, 1, 0], 60.0, (16, 16))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(1, cam.film.get_sample_bounds()), 200)
    integ.rr_enable = False
    with pytest.raises(ptrs.PtrsError, match="1024 dimensions"):
        integ.render(cam, s)
    ok = ptrs.PathIntegrator(ptrs.SamplerBuilder(1, cam.film.get_sample_bounds()), 100)
    ok.rr_enable = False
    ok.render(cam, s)
    assert ok.last_stats.rays_extension > 16 * 16 * 90


def test_cfg2_band_at_full_settings(ptrs, orc):
    """BASELINE configs[1] exactly (Cornell 1024x1024, 256 spp, depth 15) on a 16-row band = 5.3 M paths: above the 4 M-path
    threshold, so this is the SHIPPED schedule (four pipeline lanes, 2 048 segments per pass) against the oracle directly: ray
    counts identical, film rows within the tolerance of two summation orders."""
    cam, scene = ptrs.import_scene(CORNELL, (1024, 1024))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(256, cam.film.get_sample_bounds()), 15)
    rb, re = 504, 520
    integ.render(cam, scene, row_begin=rb, row_end=re)
    st = integ.last_stats
    assert st.lanes == 4 and st.queue_segments == 2048 and st.passes >= 4
    film_ref, _, ost = orc.OracleScene(scene).render(cam, orc.make_params(1024, 1024, 256, 15, row_begin=rb, row_end=re), n_threads=16)
    assert st.samples == ost.samples == 1028 * 20 * 256
    assert (st.rays_extension, st.rays_shadow, st.rays_mis) == (ost.rays_extension, ost.rays_shadow, ost.rays_mis)
    a, b = cam.film.pixels[rb:re], film_ref[rb:re]
    assert np.allclose(a["weight"], b["weight"], rtol=1e-4) and rel_l2(a["rgb"] / a["weight"][..., None], b["rgb"] / b["weight"][..., None]) < 1e-4


def test_trace_bench_and_ray_dump(ptrs, orc, scenes):
    """ptrs_render_dump_rays + ptrs_trace_bench (the traversal bench of bench.py --workload trace-*): the dumped round-0 rays are the
    camera rays of the pass, later rounds are fewer; tracing them with the frame's extension kernel gives ptrs_trace_rays' hits (which
    the oracle pins), on a quad-form scene (with the treelet node order too) and on the LDS form."""
    for order in (0, 1):
        with ptrs.options(node_order=order, lanes=1):  # (one pipeline lane: the first pass then holds all samples of the frame)
            cam, scene = scenes.triangle_soup(20000, resolution=(64, 64))
            integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(4, cam.film.get_sample_bounds()), 8)
            r0 = ptrs.dump_rays(integ, cam, scene, 0, 1 << 20)
            r2 = ptrs.dump_rays(integ, cam, scene, 2, 1 << 20)
            assert len(r0) == 68 * 68 * 4 and 0 < len(r2) < len(r0)
            assert np.isfinite(r0[:, :6]).all() and np.allclose(np.linalg.norm(r0[:, 3:6], axis=1), 1.0, atol=1e-5)
            for rays in (r0, r2):
                st, hits = ptrs.trace_bench(scene, rays, repeats=2, want_hits=True)
                ref, _ = ptrs.trace_rays(scene, rays)
                assert st.trace_launches == 2 and st.ms_trace > 0 and st.nodes_visited > 0
                assert np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["b0"].view(np.uint32), ref["b0"].view(np.uint32))
                ho, _ = orc.OracleScene(scene).trace_rays(rays)
                assert np.array_equal(hits["prim"], ho["prim"])
            if order == 1:  # the order is not a different tree: a whole render still matches the oracle
                _gpu_vs_oracle(ptrs, orc, cam, scene, 4, 8)
    cam, scene = ptrs.import_scene(CORNELL, (48, 48))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(4, cam.film.get_sample_bounds()), 6)
    rays = ptrs.dump_rays(integ, cam, scene, 1, 1 << 20)
    st, hits = ptrs.trace_bench(scene, rays, repeats=1, want_hits=True)
    ref, _ = ptrs.trace_rays(scene, rays)
    assert len(rays) > 1000 and np.array_equal(hits["prim"], ref["prim"]) and np.array_equal(hits["b2"].view(np.uint32), ref["b2"].view(np.uint32))


def test_row_cost_probe_counts_every_ray_once(ptrs):
    """ptrs_render_row_cost (the band planner's probe): one 1-spp render with a device counter per sample row.  A band [a, b) of output
    rows traces the sample rows a .. b + 3 of the grid, and output row y sits on grid row y + 2: so the BVH queries of a band render
    equal the probe's counters of output rows a - 2 .. b + 1, exactly -- for the LDS form with the fused tail (second render of the
    scene) and without, and for a quad-form scene; a second probe of the same view comes out of the cache."""
    import importlib
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    scenes = importlib.import_module("pathtracer-rs_amd.scenes")
    for make, depth in ((lambda: ptrs.import_scene(CORNELL, (96, 80)), 15), (lambda: scenes.material_zoo((72, 48)), 8)):
        cam, scene = make()
        h = cam.film.height
        cost = par.probe_row_cost(ptrs, cam, scene, depth)
        cost2 = par.probe_row_cost(ptrs, cam, scene, depth, cache=False)  # (the scene's survival profile is known now: Cornell's probe runs its late rounds in the fused tail)
        assert np.array_equal(cost, cost2) and cost.shape == (h,) and cost.min() > 0
        integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(1, cam.film.get_sample_bounds()), depth)
        for a, b in ((2, h - 2), (10, 31), (h // 2, h // 2 + 1)):
            integ.render(cam, scene, row_begin=a, row_end=b)
            assert integ.last_stats.rays == int(cost[a - 2:b + 2].astype(np.float64).sum()), (a, b)
        lib = ptrs.load_library()
        real = lib.ptrs_render_row_cost
        try:  # a second probe of the same view must not reach the library at all
            lib.ptrs_render_row_cost = None
            cost3 = par.probe_row_cost(ptrs, cam, scene, depth)
        finally:
            lib.ptrs_render_row_cost = real
        assert np.array_equal(cost, cost3)
        b8 = par.plan_bands(h, 4, cost)
        assert b8[0] == 0 and b8[-1] == h and par.plan_gain(h, 4, cost) >= 1.0 - 1e-9


def test_default_schedule_with_tail_matches_oracle_per_sample(ptrs, orc):
    """The library's DEFAULT policy on a job big enough for it (4.9 M paths: four pipeline lanes of 2 048 segments), first without a
    survival profile (every round its three launches), then with (thin rounds in the fused tail): every sample's radiance equal to
    the oracle's, bit for bit, both times -- the shipped schedule held against the oracle directly, no option set."""
    cam, scene = ptrs.import_scene(CORNELL, (192, 192))
    integ = ptrs.PathIntegrator(ptrs.SamplerBuilder(128, cam.film.get_sample_bounds()), 15)
    _, ref, ost = orc.OracleScene(scene).render(cam, orc.make_params(192, 192, 128, 15), n_threads=16, want_samples=True)
    for k in range(2):
        cam.film.clear()
        got = integ.render(cam, scene, want_samples=True)
        st = integ.last_stats
        assert st.lanes == 4 and st.queue_segments == 2048 and st.passes == 4
        assert (st.tail_launches == 0) if k == 0 else (st.tail_launches == 4 and 1 <= st.tail_round < 16)
        assert (st.rays_extension, st.rays_shadow, st.rays_mis) == (ost.rays_extension, ost.rays_shadow, ost.rays_mis)
        bad = (got.view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
        assert bad.sum() == 0, "%d of %d samples differ (render %d)" % (bad.sum(), bad.size, k)
