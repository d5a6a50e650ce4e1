// pt_texture.h -- texture evaluation: src/pathtracer/texture.rs (Constant 15-27, UVMap 29-53,
// Checker 56-89, ImageTexture::evaluate 185-191, texel 245-273, triangle 413-428,
// lookup 430-445, lookup_width 447-464).  Pyramids are built by the host (MIPMap::new).
#pragma once
#include "pt_tri.h"

namespace pt {

// Compile-time feature set of a scene (selects kernel instantiations; never changes a value):
//   FEAT_IMAGE   image textures present (MIP lookups)      FEAT_INFINITE  environment light present
//   FEAT_NORMAL  NormalMaterial wrappers present            FEAT_ALPHA     alpha-masked meshes present
enum : int { FEAT_IMAGE = 1, FEAT_INFINITE = 2, FEAT_NORMAL = 4, FEAT_ALPHA = 8, FEAT_SIMPLE = 0, FEAT_FULL = 15,
             FEAT_IMG = FEAT_IMAGE, FEAT_IMG_ENV = FEAT_IMAGE | FEAT_INFINITE }; // the two intermediate sets the shade kernels are also built for

PT_HD f3 tex_texel(const DScene &sc, const DTexture &T, uint32_t level, int32_t s, int32_t t) {
    const DTexLevel L = sc.levels[T.first_level + level];
    if (T.wrap == 0) { s = abs_mod(s, L.cols); t = abs_mod(t, L.rows); }
    else if (T.wrap == 1) { if (s < 0 || s >= L.cols || t < 0 || t >= L.rows) return splat3(0.0f); }
    else { s = s < 0 ? 0 : (s > L.cols - 1 ? L.cols - 1 : s); t = t < 0 ? 0 : (t > L.rows - 1 ? L.rows - 1 : t); }
    const float *px = sc.texdata + L.offset + ((uint64_t)t * (uint64_t)L.cols + (uint64_t)s) * (uint64_t)T.channels;
    if (T.channels == 1) return mk3(px[0], 0.0f, 0.0f);
    return mk3(px[0], px[1], px[2]);
}

PT_HD f3 tex_triangle(const DScene &sc, const DTexture &T, uint32_t level, f2 st) {
    if (level > (uint32_t)T.n_levels - 1u) level = (uint32_t)T.n_levels - 1u;
    const DTexLevel L = sc.levels[T.first_level + level];
    float s = st.x * (float)L.cols - 0.5f, t = st.y * (float)L.rows - 0.5f;
    float s0f = floor_(s), t0f = floor_(t);
    float ds = s - s0f, dt = t - t0f;
    int32_t s0 = (int32_t)s0f, t0 = (int32_t)t0f;
    return tex_texel(sc, T, level, s0, t0) * (1.0f - ds) * (1.0f - dt) + tex_texel(sc, T, level, s0, t0 + 1) * (1.0f - ds) * dt +
           tex_texel(sc, T, level, s0 + 1, t0) * ds * (1.0f - dt) + tex_texel(sc, T, level, s0 + 1, t0 + 1) * ds * dt;
}

PT_HD f3 tex_lookup_width(const DScene &sc, const DTexture &T, f2 st, float width) {
    // Footprint zero -- every lookup of a secondary ray (only the camera ray carries differentials, Q9) and every lookup of the environment map:
    // max(width, 1e-8) is 1e-8, log2 of it -26.58, and level = n_levels - 1 - 26.58 is negative for every pyramid of at most 27 levels (any image
    // below 2^27 texels a side): the reference's first branch, without the binary64 logarithm that leads there.
    if (width <= 1e-8f && T.n_levels <= 27) return tex_triangle(sc, T, 0, st);
    float nl = (float)T.n_levels;
    float level = nl - 1.0f + pt_log2f(max_nz(width, 1e-8f));
    if (level < 0.0f) return tex_triangle(sc, T, 0, st);
    if (level >= (float)(T.n_levels - 1)) return tex_triangle(sc, T, (uint32_t)T.n_levels - 1u, st);
    float il = floor_(level), delta = level - il;
    uint32_t i = (uint32_t)il;
    return lerp3(tex_triangle(sc, T, i, st), tex_triangle(sc, T, i + 1, st), delta);
}

// Texture::evaluate; 1-channel textures return their value in .x
template <int FEAT>
PT_HD f3 tex_eval(const DScene &sc, int32_t id, f2 uv, float dudx, float dvdx, float dudy, float dvdy) {
    const DTexture &T = sc.texs[id];
    if (T.kind == 0) return mk3(T.value[0], T.value[1], T.value[2]);
    f2 st = mk2(T.su * uv.x + T.du, T.sv * uv.y + T.dv);
    if (T.kind == 1) {
        float si = st.x - floor_(st.x), ti = st.y - floor_(st.y);
        bool second = (si <= 0.5f && ti <= 0.5f) || (si >= 0.5f && ti >= 0.5f);
        return second ? mk3(T.value2[0], T.value2[1], T.value2[2]) : mk3(T.value[0], T.value[1], T.value[2]);
    }
    if (!(FEAT & FEAT_IMAGE)) return splat3(0.0f); // unreachable: the scene has no image texture
    float dx0 = T.su * dudx, dx1 = T.sv * dvdx, dy0 = T.su * dudy, dy1 = T.sv * dvdy;
    float width = max_nz(max_nz(fabs_(dx0), fabs_(dx1)), max_nz(fabs_(dy0), fabs_(dy1)));
    return tex_lookup_width(sc, T, st, width);
}
template <int FEAT>
PT_HD f3 tex_eval(const DScene &sc, int32_t id, const Surface &s) { return tex_eval<FEAT>(sc, id, s.uv, s.dudx, s.dvdx, s.dudy, s.dvdy); }

} // namespace pt
