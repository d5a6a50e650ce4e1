// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Parity unpinned.
// Restates the reference's second sampler, which it compiles but never instantiates (sampler/mod.rs:169-170 aliases
// Sampler to the Sobol' one): src/pathtracer/sampler/stratified.rs:9-214 (StratifiedSamplerBuilder / StratifiedSampler),
// sampler/mod.rs:9-167 (CoreSampler, PixelSampler) and src/pathtracer/sampling.rs:5-82 (Random = rand::rngs::SmallRng,
// stratified_sample_1d/2d, shuffle, latin_hyper_cube_2d).
//
// Third-party pieces absent from /root/reference, restated from their published algorithms (Cargo.lock: rand 0.7.3,
// rand_core 0.5.1, rand_pcg 0.2.1):
//   * SmallRng on a 64-bit target = rand_pcg::Pcg64Mcg = Mcg128Xsl64: state = state * 0x2360ED051FC65DA44385DF649FCCF645
//     (mod 2^128), output = rotr64((state >> 64) ^ state, state >> 122); new(s) sets the low bit.  PINNED by the known-answer
//     vector of rand_pcg's own test (the PCG C test suite's numbers for state 42), tests/test_stratified.py.
//   * SeedableRng::seed_from_u64 (rand_core 0.5.1 default): a PCG32 stream (MUL 6364136223846793005, INC
//     11634580027462260723, XSH-RR output) fills the 16 seed bytes, little endian.  Unpinned.
//   * Rng::gen_range(0.0, 1.0) for f32 = UniformFloat::sample_single: (next_u32 >> 9) as the mantissa of a float in [1, 2),
//     minus 1, times (high - low), plus low, retried while the result is not < high.  Rng::gen::<f32>() = (next_u32 >> 8) *
//     2^-24.  Rng::gen_range(0, n) for usize = UniformInt::sample_single on u64: zone = (n << n.leading_zeros()) - 1,
//     v = next_u64, accept when the low half of v * n is <= zone, result = the high half.  next_u32 = next_u64 as u32.  Unpinned.
#pragma once
#include <cstdint>
#include <vector>

#include "orc_math.h"

namespace orc {

struct Pcg64Mcg {
    unsigned __int128 state = 1;
    static Pcg64Mcg from_state(unsigned __int128 s) { Pcg64Mcg r; r.state = s | 1; return r; }
    static Pcg64Mcg seed_from_u64(uint64_t st) { // rand_core 0.5.1 SeedableRng::seed_from_u64 + Mcg128Xsl64::from_seed
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) {
            st = st * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xorshifted = (uint32_t)(((st >> 18) ^ st) >> 27);
            const uint32_t rot = (uint32_t)(st >> 59);
            w[k] = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
        }
        const uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        return from_state((unsigned __int128)lo | ((unsigned __int128)hi << 64));
    }
    uint64_t next_u64() {
        const unsigned __int128 M = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
        state = state * M;
        const uint32_t rot = (uint32_t)(state >> 122);
        const uint64_t xsl = (uint64_t)(state >> 64) ^ (uint64_t)state;
        return (xsl >> rot) | (xsl << ((64u - rot) & 63u));
    }
    uint32_t next_u32() { return (uint32_t)next_u64(); }
    float gen_range_01() { // gen_range(0.0, 1.0): UniformFloat<f32>::sample_single
        for (;;) {
            const uint32_t bits = (next_u32() >> 9) | 0x3f800000u;
            float v12; std::memcpy(&v12, &bits, 4);
            const float res = (v12 - 1.0f) * 1.0f + 0.0f;
            if (res < 1.0f) return res;
        }
    }
    float gen_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); } // Standard: 24 bits * 2^-24
    uint64_t gen_below(uint64_t n) { // gen_range(0, n), n > 0
        const uint64_t zone = (n << __builtin_clzll(n)) - 1;
        for (;;) {
            const uint64_t v = next_u64();
            const unsigned __int128 m = (unsigned __int128)v * n;
            if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
        }
    }
};

// sampling.rs:7-18
inline void stratified_sample_1d(float *samp, size_t n, Pcg64Mcg &rng, bool jitter) {
    const float inv = 1.0f / (float)n;
    for (size_t i = 0; i < n; ++i) {
        const float delta = jitter ? rng.gen_range_01() : 0.5f;
        samp[i] = fmin_rs(((float)i + delta) * inv, ONE_MINUS_EPSILON);
    }
}
// sampling.rs:20-49
inline void stratified_sample_2d(Vec2 *samp, size_t nx, size_t ny, Pcg64Mcg &rng, bool jitter) {
    const float dx = 1.0f / (float)nx, dy = 1.0f / (float)ny;
    size_t i = 0;
    for (size_t y = 0; y < ny; ++y)
        for (size_t x = 0; x < nx; ++x) {
            const float jx = jitter ? rng.gen_range_01() : 0.5f;
            const float jy = jitter ? rng.gen_range_01() : 0.5f;
            samp[i].x = fmin_rs(((float)x + jx) * dx, ONE_MINUS_EPSILON);
            samp[i].y = fmin_rs(((float)y + jy) * dy, ONE_MINUS_EPSILON);
            ++i;
        }
}
// sampling.rs:51-59 with n_dimensions = 1
template <class T> inline void shuffle1(T *samp, size_t count, Pcg64Mcg &rng) {
    for (size_t i = 0; i < count; ++i) {
        const size_t other = i + (size_t)rng.gen_below((uint64_t)(count - i));
        T t = samp[i]; samp[i] = samp[other]; samp[other] = t;
    }
}

// StratifiedSampler with no sample arrays requested (nothing in the reference requests any).
struct StratifiedSampler {
    size_t dim_pixel_samples = 1, n_sampled_dimensions = 0, samples_per_pixel = 1;
    bool jitter_samples = true;
    Pcg64Mcg rng;
    std::vector<std::vector<float>> samples_1d;
    std::vector<std::vector<Vec2>> samples_2d;
    size_t current_1d_dimension = 0, current_2d_dimension = 0, current_pixel_sample_index = 0;
    bool drew_from_rng = false; // a draw past n_sampled_dimensions came straight from the generator (mod.rs:137-139,148-151)

    void configure(size_t dim, size_t n_dims, uint64_t seed) { // StratifiedSamplerBuilder::new + with_seed + build (stratified.rs:22-77)
        dim_pixel_samples = dim; n_sampled_dimensions = n_dims; samples_per_pixel = dim * dim;
        rng = Pcg64Mcg::seed_from_u64(seed);
        samples_1d.assign(n_dims, std::vector<float>(samples_per_pixel, 0.0f));
        Vec2 z; z.x = z.y = 0.0f;
        samples_2d.assign(n_dims, std::vector<Vec2>(samples_per_pixel, z));
    }
    void start_pixel(int32_t, int32_t) { // stratified.rs:87-148
        for (auto &v : samples_1d) { stratified_sample_1d(v.data(), samples_per_pixel, rng, jitter_samples); shuffle1(v.data(), samples_per_pixel, rng); }
        for (auto &v : samples_2d) { stratified_sample_2d(v.data(), dim_pixel_samples, dim_pixel_samples, rng, jitter_samples); shuffle1(v.data(), samples_per_pixel, rng); }
        current_pixel_sample_index = 0; // CoreSampler::start_pixel; (PixelSampler's dimension counters are NOT reset here, mod.rs:41-46: they
                                        // are zero anyway, start_next_sample having reset them at the end of the previous pixel)
    }
    bool start_next_sample() { // mod.rs:117-121, 76-81
        current_1d_dimension = current_2d_dimension = 0;
        current_pixel_sample_index += 1;
        return current_pixel_sample_index < samples_per_pixel;
    }
    float get_1d() { // mod.rs:129-141
        if (current_1d_dimension < samples_1d.size()) return samples_1d[current_1d_dimension++][current_pixel_sample_index];
        drew_from_rng = true;
        return rng.gen_range_01();
    }
    Vec2 get_2d() { // mod.rs:143-154
        if (current_2d_dimension < samples_2d.size()) return samples_2d[current_2d_dimension++][current_pixel_sample_index];
        drew_from_rng = true;
        Vec2 r; r.x = rng.gen_range_01(); r.y = rng.gen_range_01();
        return r;
    }
    Vec2 get_camera_sample(int32_t px, int32_t py) { // mod.rs:156-160
        const Vec2 u = get_2d();
        Vec2 p; p.x = (float)px + u.x; p.y = (float)py + u.y;
        return p;
    }
};

} // namespace orc
