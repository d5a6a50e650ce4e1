// pt_vec.h -- scalar / vector substrate of the device code (gfx950) and of its host twin.
//
// Everything on the hot path is binary32 arithmetic in a fixed order with no FMA contraction
// (-ffp-contract=off), IEEE division and sqrt, and the transcendental stand-ins of
// include/ptrs_detmath.h, so results are bit-identical between gfx950 and x86-64.
// Reference semantics restated here: src/common/math.rs (gamma :8, max_dimension :12-26,
// coordinate_system :48-61, next_float_up/down :71-105, offset_ray_origin :107-131,
// power_heuristic :167-171, find_interval :186-202) and nalgebra 0.32.2 vector ops.
#pragma once
#include <stdint.h>

#include "../../include/ptrs_detmath.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define PT_DEV __device__ inline
#define PT_MEM __host__ __device__ inline
#else
#define PT_DEV static inline
#define PT_MEM inline
#endif

namespace pt {

#define PT_INF (__builtin_huge_valf())
#define PT_PI 3.14159265358979323846f
#define PT_INV_PI 0.318309886183790671537767526745028724f
#define PT_PI_2 1.57079632679489661923132169163975144f
#define PT_PI_4 0.785398163397448309615660845819875721f
#define PT_INV_2PI (PT_INV_PI * 0.5f)
#define PT_ONE_MINUS_EPS 0x1.fffffep-1f
#define PT_MACH_EPS (1.1920929e-07f * 0.5f)

PT_HD float gamma_err(int n) { return ((float)n * PT_MACH_EPS) / (1.0f - (float)n * PT_MACH_EPS); }
PT_HD float fabs_(float x) { return __builtin_fabsf(x); }
PT_HD float sqrt_(float x) { return __builtin_sqrtf(x); }
PT_HD float floor_(float x) { return __builtin_floorf(x); }
PT_HD float ceil_(float x) { return __builtin_ceilf(x); }
PT_HD bool isinf_(float x) { return (ptf_bits(x) & 0x7fffffffu) == 0x7f800000u; }
PT_HD bool isnan_(float x) { return x != x; }
// Rust f32::max / f32::min (a NaN operand is ignored) and f32::clamp
// (written as three independent selects: the nested-ternary form compiles to exec-mask branches on gfx950)
PT_HD float max_(float a, float b) { const float m = a > b ? a : b; const float r = b != b ? a : m; return a != a ? b : r; }
PT_HD float min_(float a, float b) { const float m = a < b ? a : b; const float r = b != b ? a : m; return a != a ? b : r; }
// max_(0.0f, x) in two instructions: x unless it is negative or a NaN (a -0 comes back as -0, as from the comparison form)
PT_HD float max0_(float x) { return x >= 0.0f ? x : 0.0f; }
// The same where the two arguments cannot tie as +0 against -0 (absolute values, or one argument a non-zero constant): there the
// hardware's v_max_f32 / v_min_f32 (IEEE maxNum / minNum: a NaN argument yields the other one) return the same bits as the
// comparison form above in one instruction instead of six.  (With a +0 / -0 tie the comparison form returns its second argument,
// the instruction +0 for max and -0 for min: callers that can see such a tie keep max_ / min_.)
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD float max_nz(float a, float b) { return __builtin_fmaxf(a, b); }
PT_HD float min_nz(float a, float b) { return __builtin_fminf(a, b); }
#else
PT_HD float max_nz(float a, float b) { return max_(a, b); }
PT_HD float min_nz(float a, float b) { return min_(a, b); }
#endif
PT_HD float clamp_(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }

struct f2 { float x, y; };
struct f3 { float x, y, z; };
PT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD f3 splat3(float s) { return mk3(s, s, s); }
PT_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
PT_HD float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
PT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD f3 operator*(float s, f3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
// Three quotients by ONE divisor (MEASURED AND NOT USED: the default operator/ below is the compiler's division; -DPTRS_DIV3 switches it
// for A/B builds; ptrs_selftest_div3 keeps the function itself under test).  hipcc expands every IEEE binary32 division into eleven
// instructions,
//     s = v_div_scale(b, b, a); n = v_div_scale(a, b, a); r0 = v_rcp(s); e = fma(-s, r0, 1); r = fma(e, r0, r0);
//     q0 = n * r; m0 = fma(-s, q0, n); q1 = fma(m0, r, q0); m1 = fma(-s, q1, n); q = v_div_fmas(m1, r, q1); v_div_fixup(q, b, a)
// (46 cycles of a SIMD apiece, profiles/r03_valu_ceiling.json).  v_div_scale returns its operand unchanged (and VCC = 0, which makes
// v_div_fmas a plain fma) unless an exponent is extreme -- the divisor denormal or >= 2^126, the numerator below 2^-103, the
// quotient's exponent beyond +-96 / into the denormals -- and v_div_fixup returns its first operand unless an operand is zero,
// infinite or a NaN or the quotient over- / underflows.  So with |b| and every non-zero |a| in [2^-47, 2^48) (quotients in
// (2^-95, 2^95)) the expansion IS the plain chain below, whose first three instructions depend on the divisor alone: computed once,
// 3 + 3 x 5 instructions of the fast class instead of 3 x 11 with twelve slow ones.  Numerators that are zero stay on the fast path:
// the chain is run on -|b| (its values are then the exact negatives of the expansion's, rounding to nearest is symmetric) because
// with a negative divisor the signs of zero come out as IEEE's (a ^ b) for all four sign combinations -- with a positive one
// -0 / b would come back as +0 (x + (-x) = +0) -- and the quotients' sign bits are flipped afterwards where b was positive.
// Everything else (infinities, NaNs, denormals, extreme exponents) takes the compiler's division.  Identity on bits: 0 mismatches in
// 2^34 random, edge-case and render-range operand sets (tests/test_gpu_kat.py).  What it costs: the window test is 7 plain + 5
// comparison-class vector instructions and, as a divergent branch, 11 scalar ones (a SIMD issues one scalar instruction per ~4
// cycles): Cornell's Matte shade kernel 66.4 -> 69.3 ms, classroom's Disney kernels 278 -> 310 ms (ABAB).
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
PT_HD uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
PT_HD f3 div_shared3(f3 a, float s) {
    const uint32_t LO = 80u << 23, HI = 175u << 23; // biased exponents 80 .. 174: [2^-47, 2^48)
    const uint32_t bx = ptf_bits(a.x), by = ptf_bits(a.y), bz = ptf_bits(a.z), bs = ptf_bits(s);
    const uint32_t ux = bx & 0x7fffffffu, uy = by & 0x7fffffffu, uz = bz & 0x7fffffffu, us = bs & 0x7fffffffu;
    const uint32_t lo = umin_(umin_(ux - 1u, uy - 1u), uz - 1u); // v_min3_u32; a zero numerator wraps to the top and passes
    const uint32_t hi = umax_(umax_(ux, uy), uz);                // v_max3_u32; infinities and NaNs fail
    if (us - LO < HI - LO && lo >= LO - 1u && hi < HI) { // divisor in [LO, HI); numerators 0 or in [LO, HI)
        const float nb = ptf_from_bits(us | 0x80000000u); // -|b|
        const float r0 = __builtin_amdgcn_rcpf(nb);
        const float e = __builtin_fmaf(-nb, r0, 1.0f);
        const float r = __builtin_fmaf(e, r0, r0);
        const uint32_t flip = ~bs & 0x80000000u; // b > 0: the quotients by -|b| have the wrong sign
        float q[3] = {a.x, a.y, a.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float n = q[k];
            const float q0 = n * r, m0 = __builtin_fmaf(-nb, q0, n), q1 = __builtin_fmaf(m0, r, q0), m1 = __builtin_fmaf(-nb, q1, n);
            q[k] = ptf_from_bits(ptf_bits(__builtin_fmaf(m1, r, q1)) ^ flip);
        }
        return mk3(q[0], q[1], q[2]);
    }
    return mk3(a.x / s, a.y / s, a.z / s);
}
#else
PT_HD f3 div_shared3(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(PTRS_DIV3)
PT_HD f3 operator/(f3 a, float s) { return div_shared3(a, s); }
#else
PT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
#endif
PT_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PT_HD float len2(f3 a) { return dot(a, a); }
PT_HD float len(f3 a) { return sqrt_(len2(a)); }
PT_HD f3 normalize(f3 a) { return a / len(a); }
PT_HD f3 abs3(f3 a) { return mk3(fabs_(a.x), fabs_(a.y), fabs_(a.z)); }
PT_HD bool is_black(f3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
PT_HD float max_comp(f3 a) { return max_(max_(a.x, a.y), a.z); }
PT_HD float luminance(f3 c) { return c.x * 0.212671f + c.y * 0.715160f + c.z * 0.072169f; }
PT_HD f3 sqrt3(f3 a) { return mk3(sqrt_(a.x), sqrt_(a.y), sqrt_(a.z)); }
PT_HD f3 add_scalar(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
PT_HD f3 lerp3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
PT_HD f3 face_forward(f3 n, f3 v) { return dot(n, v) < 0.0f ? -n : n; }

PT_HD int max_dimension(f3 v) { return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }

PT_HD void coordinate_system(f3 v1, f3 &v2, f3 &v3) {
    if (fabs_(v1.x) > fabs_(v1.y)) v2 = mk3(-v1.z, 0.0f, v1.x) / sqrt_(v1.x * v1.x + v1.z * v1.z);
    else v2 = mk3(0.0f, v1.z, -v1.y) / sqrt_(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}

PT_HD float next_float_up(float v) {
    if (isinf_(v) && v > 0.0f) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t ui = ptf_bits(v);
    ui = (v >= 0.0f) ? ui + 1u : ui - 1u;
    return ptf_from_bits(ui);
}
// Reference quirk Q32 (math.rs:98-103): the increment/decrement branches are swapped relative to
// PBRT; reproduced as is.
PT_HD float next_float_down(float v) {
    if (isinf_(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t ui = ptf_bits(v);
    ui = (v > 0.0f) ? ui + 1u : ui - 1u;
    return ptf_from_bits(ui);
}
PT_HD float nudge(float po, float off) { return off > 0.0f ? next_float_up(po) : (off < 0.0f ? next_float_down(po) : po); }
PT_HD f3 offset_ray_origin(f3 p, f3 p_error, f3 n, f3 w) {
    float d = dot(abs3(n), p_error);
    f3 off = d * n;
    if (dot(w, n) < 0.0f) off = -off;
    f3 po = p + off;
    return mk3(nudge(po.x, off.x), nudge(po.y, off.y), nudge(po.z, off.z));
}
// A shading vertex offsets its point up to six times (light-sample pdf, shadow ray, BSDF-sample pdf, MIS ray, continuation, ...)
// with the same p, p_error and n: the direction enters only through the sign of dot(w, n).  Both outcomes once, a select per use.
struct SpawnPair { f3 n, plus, minus; };
PT_HD SpawnPair spawn_pair(f3 p, f3 p_error, f3 n) {
    SpawnPair sp; sp.n = n;
    const float d = dot(abs3(n), p_error);
    const f3 off = d * n, noff = -off;
    const f3 pp = p + off, pm = p + noff;
    sp.plus = mk3(nudge(pp.x, off.x), nudge(pp.y, off.y), nudge(pp.z, off.z));
    sp.minus = mk3(nudge(pm.x, noff.x), nudge(pm.y, noff.y), nudge(pm.z, noff.z));
    return sp;
}
PT_HD f3 spawn_from(const SpawnPair &sp, f3 w) { return dot(w, sp.n) < 0.0f ? sp.minus : sp.plus; } // == offset_ray_origin(p, p_error, n, w)
PT_HD float power_heuristic(float f_pdf, float g_pdf) { // nf = ng = 1 (1.0 * pdf is exact)
    float f = 1.0f * f_pdf, g = 1.0f * g_pdf;
    return (f * f) / (f * f + g * g);
}
PT_HD bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float &x0, float &x1) {
    float det = a00 * a11 - a01 * a10;
    if (fabs_(det) < 1e-10f) return false;
    float r0 = (a11 * b0 - a01 * b1) / det;
    float r1 = (a00 * b1 - a10 * b0) / det;
    if (r0 != r0 || r1 != r1) return false;
    x0 = r0; x1 = r1;
    return true;
}
// a mod b into [0, b) (texture.rs:246-249: `a - (a / b) * b`, plus b when negative).  For a power of two -- every level of a MIP pyramid of
// a power-of-two image -- that is `a & (b - 1)` for negative a as well (two's complement), the same integer without the ~25-instruction
// integer division; four of them per bilinear lookup.
PT_HD int32_t abs_mod(int32_t a, int32_t b) {
    if ((b & (b - 1)) == 0) return a & (b - 1);
    int32_t r = a - (a / b) * b; return r < 0 ? r + b : r;
}
// math.rs:186-202 on a cdf array: pred(i) = cdf[i] <= u
PT_HD uint32_t find_interval_cdf(const float *cdf, uint32_t size, float u) {
    uint32_t first = 0, n = size;
    while (n > 0) {
        uint32_t half = n >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; n -= half + 1; } else n = half;
    }
    uint32_t v = first - 1u; // wraps when first == 0 (Q27), then clamps to size-2
    return v > size - 2u ? size - 2u : v;
}
// The same answer with a guide table: `guide` has g+1 entries (g a power of two), guide[k] = number of leading cdf values
// <= k/g.  For u in [k/g, (k+1)/g) the count of values <= u lies in [guide[k], guide[k+1]], and bisecting that range with
// the same predicate finds it: a monotone predicate has one switch-over point, whichever way it is approached.
PT_HD uint32_t find_interval_cdf_guided(const float *cdf, uint32_t size, float u, const float *guide, uint32_t g) {
    uint32_t k = u > 0.0f ? (uint32_t)(u * (float)g) : 0u; // exact: g is a power of two
    if (k > g - 1u) k = g - 1u;
    uint32_t first = ptf_bits(guide[k]), n = ptf_bits(guide[k + 1u]) - first;
    if (u < 0.0f || !(u == u)) { first = 0; n = size; } // outside the table's domain: the plain search
    while (n > 0) {
        uint32_t half = n >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; n -= half + 1; } else n = half;
    }
    uint32_t v = first - 1u;
    return v > size - 2u ? size - 2u : v;
}
// nalgebra UnitQuaternion(i,j,k,w) * Vector3
PT_HD f3 quat_rotate(const float *q, f3 v) {
    f3 qv = mk3(q[0], q[1], q[2]);
    f3 t = cross(qv, v) * 2.0f;
    f3 c = cross(qv, t);
    return t * q[3] + c + v;
}
// row-major 3x4 affine (first 12 floats of a 4x4): Transform * Vector
PT_HD f3 xform_vec(const float *m, f3 v) {
    return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
PT_HD f3 xform_pt(const float *m, f3 p) {
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}

} // namespace pt
