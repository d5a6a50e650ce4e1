// ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's algorithm; never linked
// into, imported by or called from the product (pathtracer-rs_amd/).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Parity status: "parity unpinned" -- the reference (Rust, nightly, un-vendored crates) cannot be
// built here and its own 7 unit tests do not pin integrator results.
// The two reference known-answer tests that touch this path (common/math.rs:264-299) are
// restated in tests/test_oracle_math.py.
//
// This file: src/common/math.rs, ray.rs, bounds.rs, spectrum.rs restated.  Vector arithmetic
// follows nalgebra 0.32.2 semantics (left-to-right 3-term dot, component-wise division in
// normalize, UnitQuaternion*Vector as t = 2 q x v; v + w t + q x t).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../include/ptrs_detmath.h"

namespace orc {

static constexpr float kInf = std::numeric_limits<float>::infinity();
// math.rs:3-6
static constexpr float MACHINE_EPSILON = 1.1920929e-07f * 0.5f; // f32::EPSILON * 0.5
static constexpr float PI_F = 3.14159265358979323846f;
static constexpr float FRAC_1_PI = 0.318309886183790671537767526745028724f;
static constexpr float FRAC_PI_2 = 1.57079632679489661923132169163975144f;
static constexpr float FRAC_PI_4 = 0.785398163397448309615660845819875721f;
static constexpr float INV_2_PI = FRAC_1_PI * 0.5f;
static constexpr float ONE_MINUS_EPSILON = 0x1.fffffep-1f;
static constexpr int32_t HALF_MAX_I_32 = 2147483647 / 2;

// math.rs:8-10
inline float gamma_n(uint32_t n) { return ((float)n * MACHINE_EPSILON) / (1.0f - (float)n * MACHINE_EPSILON); }

// Rust f32::max/min (IEEE maxNum/minNum: a NaN operand is ignored)
inline float fmax_rs(float a, float b) { return a != a ? b : (b != b ? a : (a > b ? a : b)); }
inline float fmin_rs(float a, float b) { return a != a ? b : (b != b ? a : (a < b ? a : b)); }
// Rust f32::clamp: NaN stays NaN
inline float clamp_rs(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }

struct Vec2 { float x = 0, y = 0; float operator[](int i) const { return i == 0 ? x : y; } };
struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float norm_squared(Vec3 a) { return dot(a, a); }
inline float norm(Vec3 a) { return std::sqrt(norm_squared(a)); }
inline Vec3 normalize(Vec3 a) { return a / norm(a); }
inline Vec3 vabs(Vec3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline bool is_zero(Vec3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }

// math.rs:12-26
inline int max_dimension(Vec3 v) { return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }
inline Vec3 permute(Vec3 p, int x, int y, int z) { return {p[x], p[y], p[z]}; }
// math.rs:37-46
inline Vec3 face_forward(Vec3 n, Vec3 v) { return dot(n, v) < 0.0f ? -n : n; }
// math.rs:48-61
inline void coordinate_system(Vec3 v1, Vec3 &v2, Vec3 &v3) {
    if (std::fabs(v1.x) > std::fabs(v1.y)) v2 = Vec3(-v1.z, 0.0f, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else v2 = Vec3(0.0f, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    v3 = cross(v1, v2);
}
inline uint32_t float_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
// math.rs:71-105
inline float next_float_up(float v) {
    if (std::isinf(v) && v > 0.0f) return v;
    if (v == -0.0f) v = 0.0f;
    uint32_t ui = float_to_bits(v);
    if (v >= 0.0f) ui += 1; else ui -= 1;
    return bits_to_float(ui);
}
inline float next_float_down(float v) {
    if (std::isinf(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t ui = float_to_bits(v);
    // Q32 (reference quirk, math.rs:98-103): the branches are swapped relative to PBRT, so a
    // positive v moves UP one ulp, a negative v moves toward zero, and -0.0 becomes 0x7fffffff (NaN).
    if (v > 0.0f) ui += 1; else ui -= 1;
    return bits_to_float(ui);
}
// math.rs:107-131
inline Vec3 offset_ray_origin(Vec3 p, Vec3 p_error, Vec3 n, Vec3 w) {
    float d = dot(vabs(n), p_error);
    Vec3 offset = d * n;
    if (dot(w, n) < 0.0f) offset = -offset;
    Vec3 po = p + offset;
    for (int i = 0; i < 3; i++) {
        if (offset[i] > 0.0f) po[i] = next_float_up(po[i]);
        else if (offset[i] < 0.0f) po[i] = next_float_down(po[i]);
    }
    return po;
}
// math.rs:133-147
inline float gamma_correct(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * pt_powf(v, 1.0f / 2.4f) - 0.055f; }
inline float inverse_gamma_correct(float v) { return v <= 0.04045f ? v * 1.0f / 12.92f : pt_powf((v + 0.055f) * 1.0f / 1.055f, 2.4f); }
// math.rs:149-165 (a is row-major a00 a01 a10 a11)
inline bool solve_linear_system_2x2(const float a[4], const float b[2], float x[2]) {
    float det = a[0] * a[3] - a[1] * a[2]; // nalgebra Matrix2::determinant: m11*m22 - m21*m12
    if (std::fabs(det) < 1e-10f) return false;
    float x0 = (a[3] * b[0] - a[1] * b[1]) / det;
    float x1 = (a[0] * b[1] - a[2] * b[0]) / det;
    if (x0 != x0 || x1 != x1) return false;
    x[0] = x0; x[1] = x1;
    return true;
}
// math.rs:167-171
inline float power_heuristic(int nf, float f_pdf, int ng, float g_pdf) {
    float f = (float)nf * f_pdf, g = (float)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}
// math.rs:173-184
inline float spherical_theta(Vec3 v) { return pt_acosf(clamp_rs(v.z, -1.0f, 1.0f)); }
inline float spherical_phi(Vec3 v) { float p = pt_atan2f(v.y, v.x); return p < 0.0f ? p + 2.0f * PI_F : p; }
// math.rs:186-202; pred(i) must be callable with i in [0,size)
template <class P> inline size_t find_interval(size_t size, P pred) {
    size_t first = 0, len = size;
    while (len > 0) {
        size_t half = len >> 1, middle = first + half;
        if (pred(middle)) { first = middle + 1; len -= half + 1; } else { len = half; }
    }
    size_t v = first - 1; // usize wrap when first == 0 (Q27)
    size_t hi = size - 2;
    return v > hi ? hi : v; // clamp(0, size-2) on usize
}
// math.rs:204-232
inline int32_t round_up_pow2_i32(int32_t v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline int64_t round_up_pow2_i64(int64_t v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; return v + 1; }
// math.rs:234-241
inline int32_t abs_mod(int32_t a, int32_t b) { int32_t r = a - (a / b) * b; return r < 0 ? r + b : r; }
// math.rs:243-245 (usize = 64 bit)
inline uint32_t log2_int(uint64_t i) { return 63u - (uint32_t)__builtin_clzll(i); }
// math.rs:247-254
template <class T> inline T lerp(T x, T y, float a) { return x * (1.0f - a) + y * a; }
// math.rs:256-258 (usize arithmetic, wrapping irrelevant for the magnitudes involved)
inline uint64_t cantor_pairing(uint64_t x, uint64_t y) { return (x + y) * (x + y + 1) / 2 + y; }

// ---- spectrum.rs ------------------------------------------------------------------------------
struct Spectrum {
    float r = 0, g = 0, b = 0;
    Spectrum() = default;
    explicit Spectrum(float c) : r(c), g(c), b(c) {}
    Spectrum(float a, float b_, float c) : r(a), g(b_), b(c) {}
    bool is_black() const { return r == 0.0f && g == 0.0f && b == 0.0f; }
    bool has_nan() const { return r != r || g != g || b != b; }
    float y() const { return r * 0.212671f + g * 0.715160f + b * 0.072169f; }
    float max_component_value() const { return fmax_rs(fmax_rs(r, g), b); }
    float operator[](int i) const { return i == 0 ? r : (i == 1 ? g : b); }
};
inline Spectrum operator+(Spectrum a, Spectrum b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Spectrum operator-(Spectrum a, Spectrum b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Spectrum operator-(Spectrum a, float s) { return {a.r + (-s), a.g + (-s), a.b + (-s)}; } // add_scalar(-rhs)
inline Spectrum operator*(Spectrum a, Spectrum b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Spectrum operator*(Spectrum a, float s) { return {a.r * s, a.g * s, a.b * s}; }
inline Spectrum operator*(float s, Spectrum a) { return a * s; }
inline Spectrum operator/(Spectrum a, Spectrum b) { return {a.r / b.r, a.g / b.g, a.b / b.b}; }
inline Spectrum operator/(Spectrum a, float s) { return {a.r / s, a.g / s, a.b / s}; }
inline Spectrum &operator+=(Spectrum &a, Spectrum b) { a = a + b; return a; }
inline Spectrum &operator*=(Spectrum &a, Spectrum b) { a = a * b; return a; }
inline Spectrum &operator*=(Spectrum &a, float s) { a = a * s; return a; }
inline Spectrum &operator/=(Spectrum &a, float s) { a = a / s; return a; }
inline Spectrum sqrt(Spectrum a) { return {std::sqrt(a.r), std::sqrt(a.g), std::sqrt(a.b)}; }

// ---- ray.rs -----------------------------------------------------------------------------------
struct Ray { Vec3 o, d; float t_max = kInf; };
struct RayDifferential {
    Ray ray;
    bool has_differentials = false;
    Vec3 rx_origin, ry_origin, rx_direction, ry_direction;
    RayDifferential() = default;
    explicit RayDifferential(const Ray &r) : ray(r) {}
    // ray.rs:30-35
    void scale_differentials(float s) {
        rx_origin = ray.o + (rx_origin - ray.o) * s;
        ry_origin = ray.o + (ry_origin - ray.o) * s;
        rx_direction = ray.d + (rx_direction - ray.d) * s;
        ry_direction = ray.d + (ry_direction - ray.d) * s;
    }
};

// ---- bounds.rs --------------------------------------------------------------------------------
struct Bounds3 {
    Vec3 p_min, p_max;
    // bounds.rs:79-88: num::Bounded min/max = +-f32::MAX
    static Bounds3 empty() {
        float mx = std::numeric_limits<float>::max();
        Bounds3 b; b.p_min = Vec3(mx, mx, mx); b.p_max = Vec3(-mx, -mx, -mx); return b;
    }
    static Vec3 min_p(Vec3 a, Vec3 b) { return {fmin_rs(a.x, b.x), fmin_rs(a.y, b.y), fmin_rs(a.z, b.z)}; }
    static Vec3 max_p(Vec3 a, Vec3 b) { return {fmax_rs(a.x, b.x), fmax_rs(a.y, b.y), fmax_rs(a.z, b.z)}; }
    static Bounds3 from_points(Vec3 a, Vec3 b) { Bounds3 r; r.p_min = min_p(a, b); r.p_max = max_p(a, b); return r; }
    static Bounds3 union_b(const Bounds3 &a, const Bounds3 &b) { Bounds3 r; r.p_min = min_p(a.p_min, b.p_min); r.p_max = max_p(a.p_max, b.p_max); return r; }
    static Bounds3 union_p(const Bounds3 &a, Vec3 p) { Bounds3 r; r.p_min = min_p(a.p_min, p); r.p_max = max_p(a.p_max, p); return r; }
    Vec3 diagonal() const { return p_max - p_min; }
    // bounds.rs:93-95, nalgebra imax: first index of the maximum
    int maximum_extent() const { Vec3 d = diagonal(); int i = 0; float m = d.x; if (d.y > m) { m = d.y; i = 1; } if (d.z > m) { i = 2; } return i; }
    // bounds.rs:97-110
    Vec3 offset(Vec3 p) const {
        Vec3 o = p - p_min;
        if (p_max.x > p_min.x) o.x /= p_max.x - p_min.x;
        if (p_max.y > p_min.y) o.y /= p_max.y - p_min.y;
        if (p_max.z > p_min.z) o.z /= p_max.z - p_min.z;
        return o;
    }
    float surface_area() const { Vec3 d = diagonal(); return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z); }
    static bool inside(Vec3 p, const Bounds3 &b) { return p.x >= b.p_min.x && p.x <= b.p_max.x && p.y >= b.p_min.y && p.y <= b.p_max.y && p.z >= b.p_min.z && p.z <= b.p_max.z; }
    // bounds.rs:126-134
    void bounding_sphere(Vec3 &center, float &radius) const {
        center = (p_min + p_max) * 0.5f;
        radius = inside(center, *this) ? norm(center - p_max) : 0.0f;
    }
    const Vec3 &operator[](bool i) const { return i ? p_max : p_min; }
    // bounds.rs:190-232
    bool intersect_p_precomp(const Ray &r, Vec3 inv_dir, const bool dir_is_neg[3]) const {
        const Bounds3 &s = *this;
        float t_min = (s[dir_is_neg[0]].x - r.o.x) * inv_dir.x;
        float t_max = (s[!dir_is_neg[0]].x - r.o.x) * inv_dir.x;
        float ty_min = (s[dir_is_neg[1]].y - r.o.y) * inv_dir.y;
        float ty_max = (s[!dir_is_neg[1]].y - r.o.y) * inv_dir.y;
        t_max *= 1.0f + 2.0f * gamma_n(3);
        ty_max *= 1.0f + 2.0f * gamma_n(3);
        if (t_min > ty_max || ty_min > t_max) return false;
        if (ty_min > t_min) t_min = ty_min;
        if (ty_max < t_max) t_max = ty_max;
        float tz_min = (s[dir_is_neg[2]].z - r.o.z) * inv_dir.z;
        float tz_max = (s[!dir_is_neg[2]].z - r.o.z) * inv_dir.z;
        tz_max *= 1.0f + 2.0f * gamma_n(3);
        if (t_min > tz_max || tz_min > t_max) return false;
        if (tz_min > t_min) t_min = tz_min;
        if (tz_max < t_max) t_max = tz_max;
        return (t_min < r.t_max) && (t_max > 0.0f);
    }
};

// nalgebra UnitQuaternion (i,j,k,w) * Vector3
inline Vec3 quat_rotate(const float q[4], Vec3 v) {
    Vec3 qv(q[0], q[1], q[2]);
    Vec3 t = cross(qv, v) * 2.0f;
    Vec3 c = cross(qv, t);
    return t * q[3] + c + v;
}
// nalgebra Transform(4x4 row-major) * Point3 for an affine matrix: ((m0*x + m1*y) + m2*z) + t
inline Vec3 affine_point(const float m[16], Vec3 p) {
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
inline Vec3 affine_vector(const float m[16], Vec3 v) {
    return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

} // namespace orc
