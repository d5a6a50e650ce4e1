// pt_sobol.h -- device Sobol' sampler: src/pathtracer/sampler/sobol.rs:81-193 and
// lowdiscrepancy.rs:9-57 for a sampler with no sample arrays (array_end_dim == ARRAY_START_DIM == 5).
#pragma once
#include "pt_scene.h"

namespace pt {

// sobol_interval_to_index (lowdiscrepancy.rs:9-39); px,py relative to the sample bounds
PT_HD uint64_t sobol_index(const DSampler &S, uint64_t frame, uint32_t px, uint32_t py) {
    const uint32_t m = S.log2_res;
    if (m == 0) return 0;
    uint64_t index = frame << (m << 1);
    uint64_t delta = 0;
    for (uint32_t c = 0; frame != 0; frame >>= 1, ++c)
        if (frame & 1) delta ^= S.vdc[c];
    uint64_t b = ((uint64_t)(px << m) | (uint64_t)py) ^ delta;
    for (uint32_t c = 0; b != 0; b >>= 1, ++c)
        if (b & 1) index ^= S.vdc_inv[c];
    return index;
}

// cantor_pairing(x + HALF_MAX, y + HALF_MAX) truncated to 32 bits (sobol.rs:83-86, math.rs:256-258)
PT_HD uint32_t pixel_scramble(int32_t x, int32_t y) {
    uint64_t a = (uint64_t)(int64_t)(x + 1073741823), b = (uint64_t)(int64_t)(y + 1073741823);
    return (uint32_t)((a + b) * (a + b + 1) / 2 + b);
}

// sobol_sample (lowdiscrepancy.rs:42-57).  v = scramble XOR (XOR of matrix columns at the set bits
// of index).  XOR is associative, so the columns are pre-combined per index byte
// (bytetab[dim][k][b] = XOR_{j in bits(b)} M[dim][8k+j]); the result is bit-identical to the
// bit-serial loop, which remains as the fallback for dimensions beyond the table.
PT_HD float sobol_sample(const DSampler &S, uint64_t index, uint32_t dim, uint32_t scramble) {
    uint32_t v = scramble;
    if (S.bytetab && dim < (uint32_t)SOBOL_TAB_DIMS) {
        const uint32_t *t = S.bytetab + (size_t)dim * (8u * 256u);
        uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
        v ^= t[lo & 255u] ^ t[256u + ((lo >> 8) & 255u)] ^ t[512u + ((lo >> 16) & 255u)] ^ t[768u + (lo >> 24)];
        if (hi) v ^= t[1024u + (hi & 255u)] ^ t[1280u + ((hi >> 8) & 255u)] ^ t[1536u + ((hi >> 16) & 255u)] ^ t[1792u + (hi >> 24)];
    } else {
        const uint32_t *mat = S.matrices + dim * 52u;
        for (; index != 0; index >>= 1, ++mat)
            if (index & 1) v ^= *mat;
    }
    return min_nz(PT_ONE_MINUS_EPS, (float)v * 0x1p-32f);
}

// sample_dimension (sobol.rs:177-193): dimensions 0/1 are mapped to the pixel and clamped (Q1)
PT_HD float sample_dimension(const DSampler &S, uint64_t index, uint32_t dim, uint32_t scramble, int32_t px, int32_t py) {
    float s = sobol_sample(S, index, dim, scramble);
    if (dim < 2) {
        s = s * (float)S.resolution + (float)(dim == 0 ? S.min_x : S.min_y);
        s = clamp_(s - (float)(dim == 0 ? px : py), 0.0f, PT_ONE_MINUS_EPS);
    }
    return s;
}

// N dimensions >= 2 of one index at once: the same values as N sobol_sample calls, with all table reads issued before the
// first one is needed (one memory round trip per shading vertex instead of one per dimension).
template <int N>
PT_HD void sobol_batch(const DSampler &S, uint64_t index, const uint32_t (&dim)[N], uint32_t scramble, float (&out)[N]) {
    const uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
    uint32_t v[N];
#ifdef PTRS_ABLATE_SOBOL // diagnostic builds only (tools/ablate.sh): no table reads, wrong values
    for (int k = 0; k < N; ++k) { uint32_t w = (lo ^ scramble) * 2654435761u + dim[k] * 40503u; w ^= w >> 15; out[k] = min_(PT_ONE_MINUS_EPS, (float)(w * 2246822519u) * 0x1p-32f); }
    return;
#endif
    bool tab = S.bytetab != nullptr;
#pragma unroll
    for (int k = 0; k < N; ++k) tab = tab && dim[k] < (uint32_t)SOBOL_TAB_DIMS;
    if (tab) {
        uint32_t a[N], b[N], c[N], d[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const uint32_t *t = S.bytetab + (size_t)dim[k] * (8u * 256u);
            a[k] = t[lo & 255u]; b[k] = t[256u + ((lo >> 8) & 255u)]; c[k] = t[512u + ((lo >> 16) & 255u)]; d[k] = t[768u + (lo >> 24)];
        }
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = scramble ^ a[k] ^ b[k] ^ c[k] ^ d[k];
        if (hi) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const uint32_t *t = S.bytetab + (size_t)dim[k] * (8u * 256u);
                a[k] = t[1024u + (hi & 255u)]; b[k] = t[1280u + ((hi >> 8) & 255u)]; c[k] = t[1536u + ((hi >> 16) & 255u)]; d[k] = t[1792u + (hi >> 24)];
            }
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] ^= a[k] ^ b[k] ^ c[k] ^ d[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            uint32_t w = scramble;
            const uint32_t *mat = S.matrices + dim[k] * 52u;
            for (uint64_t i = index; i != 0; i >>= 1, ++mat)
                if (i & 1) w ^= *mat;
            v[k] = w;
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) out[k] = min_nz(PT_ONE_MINUS_EPS, (float)v[k] * 0x1p-32f);
}

// The dimensions one shading vertex draws, in the reference's order (integrator.rs:202-216, 453, 491): with next-event
// estimation u_light (2-D), u_scatter (2-D), the light choice (1-D); then the BSDF sample (2-D) and the Russian-roulette
// sample (1-D, drawn only inside its test).  get_2d's rule (dimension 4 is skipped, Q2) is applied to every 2-D draw.
struct VertexDims { uint32_t nee[5], cont[2], rr; uint32_t after_nee, after_cont; };
PT_HD VertexDims vertex_dims(uint32_t dim, bool nee) {
    VertexDims V;
    if (nee) {
        if (dim == 4) dim = 5;
        V.nee[0] = dim; V.nee[1] = dim + 1; dim += 2;
        if (dim == 4) dim = 5;
        V.nee[2] = dim; V.nee[3] = dim + 1; dim += 2;
        V.nee[4] = dim; dim += 1;
    } else { V.nee[0] = V.nee[1] = V.nee[2] = V.nee[3] = V.nee[4] = 2; }
    V.after_nee = dim;
    if (dim == 4) dim = 5;
    V.cont[0] = dim; V.cont[1] = dim + 1; dim += 2;
    V.after_cont = dim;
    V.rr = dim;
    return V;
}

// running sampler state of one path: index, dimension counter, pixel scramble
struct SamplerState {
    uint64_t index;
    uint32_t dim;
    uint32_t scramble;
    int32_t px, py;
};
PT_HD float get_1d(const DSampler &S, SamplerState &st) { // sobol.rs:129-137 (the jump can never trigger)
    float s = sample_dimension(S, st.index, st.dim, st.scramble, st.px, st.py);
    st.dim += 1;
    return s;
}
PT_HD f2 get_2d(const DSampler &S, SamplerState &st) { // sobol.rs:139-151: dimension 4 is skipped (Q2)
    if (st.dim == 4) st.dim = 5;
    f2 r;
    r.x = sample_dimension(S, st.index, st.dim, st.scramble, st.px, st.py);
    r.y = sample_dimension(S, st.index, st.dim + 1, st.scramble, st.px, st.py);
    st.dim += 2;
    return r;
}

} // namespace pt
