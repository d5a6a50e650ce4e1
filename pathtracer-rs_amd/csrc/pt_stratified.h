// pt_stratified.h -- the reference's second sampler on the device: StratifiedSampler (src/pathtracer/sampler/stratified.rs:87-148,
// sampler/mod.rs:94-167) with its generator, Random = rand::rngs::SmallRng (src/pathtracer/sampling.rs:5-59).
//
// The reference compiles this sampler but never builds one (sampler/mod.rs:169-170); here it is a second DSampler kind
// (PtrsRenderParams.sampler).  Its per-pixel tables come out of ONE sequential generator per 16x16 tile, re-seeded with the
// tile's index (integrator.rs:553-554) and consumed in the tile's pixel order (x outer, y inner), so the tables of a frame
// are made by one GPU thread per tile (k_strat_tables) before the passes start; a path then reads table entries where the
// Sobol' sampler would compute them.  A draw past n_sampled_dimensions would come straight from that generator in the order
// the tile's paths happen to run (mod.rs:137-151) -- not reproducible by a wavefront, refused by render_impl.
//
// rand 0.7.3 / rand_core 0.5.1 / rand_pcg 0.2.1 are not in /root/reference: restated from their published algorithms, see
// oracle/orc_stratified.h for the list.  Parity unpinned except Pcg64Mcg's output function (pinned by rand_pcg's known-answer vector).
#pragma once
#include "pt_scene.h"

namespace pt {

PT_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

struct Pcg64Mcg { // Mcg128Xsl64: 128-bit multiplicative congruential state, XSL-RR output
    uint64_t lo, hi;
    PT_MEM uint64_t next_u64() {
        const uint64_t M_LO = 0x4385DF649FCCF645ull, M_HI = 0x2360ED051FC65DA4ull;
        const uint64_t nlo = lo * M_LO;
        const uint64_t nhi = mulhi64(lo, M_LO) + lo * M_HI + hi * M_LO;
        lo = nlo; hi = nhi;
        const uint32_t rot = (uint32_t)(hi >> 58);
        const uint64_t xsl = hi ^ lo;
        return (xsl >> rot) | (xsl << ((64u - rot) & 63u));
    }
    PT_MEM uint32_t next_u32() { return (uint32_t)next_u64(); }
    PT_MEM float gen_range_01() { // Rng::gen_range(0.0, 1.0): UniformFloat<f32>::sample_single
        for (;;) {
            const float v12 = u2f((next_u32() >> 9) | 0x3f800000u);
            const float res = (v12 - 1.0f) * 1.0f + 0.0f;
            if (res < 1.0f) return res;
        }
    }
    PT_MEM uint64_t gen_below(uint64_t n) { // Rng::gen_range(0, n): UniformInt<usize>::sample_single
        const uint64_t zone = (n << __builtin_clzll(n)) - 1;
        for (;;) {
            const uint64_t v = next_u64();
            if (v * n <= zone) return mulhi64(v, n);
        }
    }
};
PT_HD Pcg64Mcg pcg_seed_from_u64(uint64_t st) { // SeedableRng::seed_from_u64 (rand_core 0.5.1) + Mcg128Xsl64::from_seed / new
    uint32_t w[4];
    for (int k = 0; k < 4; ++k) {
        st = st * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((st >> 18) ^ st) >> 27);
        const uint32_t rot = (uint32_t)(st >> 59);
        w[k] = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    Pcg64Mcg r;
    r.lo = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) | 1ull;
    r.hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    return r;
}

// One tile's tables: for each pixel in (x outer, y inner) order stratified_sample_1d + shuffle for every 1-D dimension, then
// stratified_sample_2d + shuffle for every 2-D dimension (stratified.rs:87-116; sampling.rs:7-59), jitter on.
// tab1[(pixel * n_dims + d) * spp + s], tab2[((pixel * n_dims + d) * spp + s) * 2 + {0,1}], pixel = sy * NX + sx of the sample grid.
PT_HD void stratified_tile_tables(uint64_t seed, int32_t x0, int32_t x1, int32_t y0, int32_t y1, int32_t NX, uint32_t dim_ps, uint32_t n_dims, float *tab1, float *tab2) {
    Pcg64Mcg rng = pcg_seed_from_u64(seed);
    const uint32_t spp = dim_ps * dim_ps;
    const float inv_n = 1.0f / (float)spp, dxy = 1.0f / (float)dim_ps;
    for (int32_t x = x0; x < x1; ++x)
        for (int32_t y = y0; y < y1; ++y) {
            const size_t pixel = (size_t)y * (size_t)NX + (size_t)x;
            for (uint32_t d = 0; d < n_dims; ++d) {
                float *s1 = tab1 + (pixel * n_dims + d) * spp;
                for (uint32_t i = 0; i < spp; ++i) s1[i] = min_(((float)i + rng.gen_range_01()) * inv_n, PT_ONE_MINUS_EPS);
                for (uint32_t i = 0; i < spp; ++i) { const uint32_t o = i + (uint32_t)rng.gen_below((uint64_t)(spp - i)); const float t = s1[i]; s1[i] = s1[o]; s1[o] = t; }
            }
            for (uint32_t d = 0; d < n_dims; ++d) {
                float *s2 = tab2 + (pixel * n_dims + d) * spp * 2;
                uint32_t i = 0;
                for (uint32_t yy = 0; yy < dim_ps; ++yy)
                    for (uint32_t xx = 0; xx < dim_ps; ++xx, ++i) {
                        const float jx = rng.gen_range_01(), jy = rng.gen_range_01();
                        s2[2 * i] = min_(((float)xx + jx) * dxy, PT_ONE_MINUS_EPS);
                        s2[2 * i + 1] = min_(((float)yy + jy) * dxy, PT_ONE_MINUS_EPS);
                    }
                for (uint32_t k = 0; k < spp; ++k) {
                    const uint32_t o = k + (uint32_t)rng.gen_below((uint64_t)(spp - k));
                    const float tx = s2[2 * k], ty = s2[2 * k + 1]; s2[2 * k] = s2[2 * o]; s2[2 * k + 1] = s2[2 * o + 1]; s2[2 * o] = tx; s2[2 * o + 1] = ty;
                }
            }
        }
}

} // namespace pt
