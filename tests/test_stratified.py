"""The reference's second sampler (src/pathtracer/sampler/stratified.rs, never instantiated there: sampler/mod.rs:169-170)
as PtrsRenderParams.sampler = PTRS_SAMPLER_STRATIFIED.  Its generator, rand 0.7.3's SmallRng = rand_pcg 0.2.1's Pcg64Mcg,
is not in /root/reference: restated from the published algorithm and PINNED here by rand_pcg's own known-answer vector;
seed_from_u64 and the gen_range mappings are restated too and stay unpinned (see oracle/orc_stratified.h)."""
import numpy as np
import pytest

import twin
from conftest import CORNELL


def test_pcg64mcg_known_answer(orc):
    """rand_pcg 0.2.1, pcg128.rs test `test_mcg128xsl64_true_values`: Mcg128Xsl64::new(42), six outputs -- the numbers of the
    PCG C test suite."""
    got = orc.pcg64mcg(42, 6)
    want = [0x63b4a3a813ce700a, 0x382954200617ab24, 0xa7fd85ae3fe950ce, 0xd715286aa2887737, 0x60c92fee2e59f32c, 0x84c4e96beff30017]
    assert [int(v) for v in got] == want


def test_stratified_tables_are_stratified(orc):
    """stratified_sample_1d / _2d + shuffle (sampling.rs:7-59): every stratum of every dimension holds exactly one sample."""
    dim, nd = 4, 5
    spp = dim * dim
    t = orc.stratified_tile(3, 3, 2, dim, nd)  # 6 pixels
    for px in range(6):
        s1 = t[px, : nd * spp].reshape(nd, spp)
        s2 = t[px, nd * spp:].reshape(nd, spp, 2)
        assert (s1 >= 0).all() and (s1 < 1).all() and (s2 >= 0).all() and (s2 < 1).all()
        for d in range(nd):
            assert sorted(np.floor(s1[d] * spp).astype(int)) == list(range(spp))
            cells = np.floor(s2[d] * dim).astype(int)
            assert sorted(cells[:, 1] * dim + cells[:, 0]) == list(range(spp))
    # another seed, other tables; the same seed, the same tables
    assert not np.array_equal(t, orc.stratified_tile(4, 3, 2, dim, nd))
    assert np.array_equal(t, orc.stratified_tile(3, 3, 2, dim, nd))


@pytest.mark.parametrize("res,dim,depth", [((40, 36), 2, 4), ((21, 33), 3, 2)])
def test_twin_matches_oracle_with_the_stratified_sampler(ptrs, orc, res, dim, depth):
    """The product's device code on the CPU (host twin: tables by tile, table reads instead of Sobol' evaluations) against the
    oracle's StratifiedSampler driven like the reference's tile loop (integrator.rs:551-611): every sample bit-identical."""
    cam, scene = ptrs.import_scene(CORNELL, res)
    p = orc.make_params(res[0], res[1], dim * dim, depth, sampler=ptrs.abi.SAMPLER_STRATIFIED, n_sampled_dimensions=3 * (depth + 1) + 1)
    fo, so, sto = orc.OracleScene(scene).render(cam, p, n_threads=4, want_samples=True)
    ft, stw, stt = twin.TwinScene(scene).render(cam, p, want_samples=True)
    assert sto.error_flags == 0
    assert (stt.samples, stt.rays_extension, stt.rays_shadow, stt.rays_mis) == (sto.samples, sto.rays_extension, sto.rays_shadow, sto.rays_mis)
    assert np.array_equal(so.view(np.uint32), stw.view(np.uint32))
    # a band sees the same samples: the generator runs through whole tiles whichever rows are rendered
    pb = orc.make_params(res[0], res[1], dim * dim, depth, row_begin=10, row_end=20, sampler=ptrs.abi.SAMPLER_STRATIFIED, n_sampled_dimensions=3 * (depth + 1) + 1)
    fb, sb, _ = twin.TwinScene(scene).render(cam, pb, want_samples=True)
    assert np.array_equal(sb[10:22].view(np.uint32), stw[10:22].view(np.uint32))
    assert np.array_equal(fb["rgb"][10:20].view(np.uint32), ft["rgb"][10:20].view(np.uint32))


def test_stratified_parameter_errors(ptrs, orc):
    cam, scene = ptrs.import_scene(CORNELL, (16, 16))
    t = twin.TwinScene(scene)
    for spp, nd, depth in ((8, 40, 4), (9, 5, 4), (9, 64, 4)):  # not a square; too few dimensions for the depth; more than the state word holds
        p = orc.make_params(16, 16, spp, depth, sampler=ptrs.abi.SAMPLER_STRATIFIED, n_sampled_dimensions=nd)
        with pytest.raises(RuntimeError):
            t.render(cam, p)


@pytest.mark.gpu
def test_gpu_matches_oracle_with_the_stratified_sampler(ptrs, orc, scenes):
    """k_strat_tables (one GPU thread per tile: the 128-bit generator, jitter, shuffles) + the table-reading shade kernels
    against the oracle: per-sample radiance bit-identical, on Cornell and on the material zoo (specular chains, glass)."""
    # (the zoo's glass yields null-BSDF skips, which hand a path extra vertices (Q7): it needs more dimensions than 3 (depth + 1) + 1)
    for cam, scene, dim, depth, nd in ((ptrs.import_scene(CORNELL, (64, 48)) + (3, 5, 19)), (scenes.material_zoo((60, 40)) + (2, 5, 40))):
        integ = ptrs.PathIntegrator(ptrs.StratifiedSamplerBuilder(dim, nd), depth)
        got = integ.render(cam, scene, want_samples=True)
        st = integ.last_stats
        p = orc.make_params(cam.film.width, cam.film.height, dim * dim, depth, sampler=ptrs.abi.SAMPLER_STRATIFIED, n_sampled_dimensions=nd)
        fo, so, sto = orc.OracleScene(scene).render(cam, p, n_threads=8, want_samples=True)
        assert (st.samples, st.rays_extension, st.rays_shadow, st.rays_mis) == (sto.samples, sto.rays_extension, sto.rays_shadow, sto.rays_mis)
        assert sto.error_flags == 0
        assert np.array_equal(got.view(np.uint32), so.view(np.uint32))
    # too few dimensions for the zoo's longest paths: the oracle's sampler falls back to its generator, the GPU render refuses
    with pytest.raises(ptrs.PtrsError, match="n_sampled_dimensions"):
        ptrs.PathIntegrator(ptrs.StratifiedSamplerBuilder(2, 19), 5).render(cam, scene)
    with pytest.raises(ptrs.PtrsError, match="n_sampled_dimensions"):
        ptrs.PathIntegrator(ptrs.StratifiedSamplerBuilder(2, 10), 15).render(cam, scene)
