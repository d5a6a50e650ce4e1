// pt_tri.h -- watertight ray/triangle test and hit-record reconstruction.
//
// Restates src/pathtracer/shape.rs: Triangle::intersect (74-360) split in two:
//   tri_test()     the accept/reject part (80-185 and, identically, intersect_p 362-468):
//                  runs once per candidate inside BVH traversal, produces (t, b0, b1, b2);
//   tri_surface()  the hit-record part (187-356): runs once per accepted closest hit in the
//                  shade stage from (triangle, b0, b1, b2, wo).  It is a pure function of those,
//                  so deferring it does not change any value.
// Degenerate-triangle rejection (205-212) depends only on the triangle and is folded into
// tri_test through tri_degenerate().  dndu/dndv (313-351) feed only dead code in the reference
// (integrator.rs:264-390) and are not computed.
#pragma once
#include "pt_scene.h"

namespace pt {

struct TriHit { float t, b0, b1, b2; };

// The per-ray part of Triangle::intersect (shape.rs:92-109): the permutation that makes |d| largest in z and the
// shear that aligns d with +z.  It depends on the ray only, so traversal computes it once instead of once per triangle
// (three IEEE divisions each time).
struct RayShear { int kz; float sx, sy, sz; };
PT_HD RayShear ray_shear(f3 d) {
    RayShear S;
    S.kz = max_dimension(abs3(d));
    int kx = S.kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    const float dx = comp(d, kx), dy = comp(d, ky), dz = comp(d, S.kz);
    S.sx = -dx / dz; S.sy = -dy / dz; S.sz = 1.0f / dz;
    return S;
}

// The test proper on vertices that are already translated to the ray origin and permuted to (kx, ky, kz) -- shape.rs:110-185.
// (Callers that keep permuted copies of the triangles -- the LDS-form traversal kernels -- skip the nine component selects.)
PT_HD bool tri_test_perm(f3 p0t, f3 p1t, f3 p2t, float sx, float sy, float sz, float t_max, TriHit &h) {
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { // binary64 fallback, shape.rs:124-134
        e0 = (float)((double)p2t.y * (double)p1t.x - (double)p2t.x * (double)p1t.y);
        e1 = (float)((double)p0t.y * (double)p2t.x - (double)p0t.x * (double)p2t.y);
        e2 = (float)((double)p1t.y * (double)p0t.x - (double)p1t.x * (double)p0t.y);
    }
    const float det = e0 + e1 + e2;
    // the common rejection (edge signs disagree) and the degenerate case in one test
    if ((((e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f)) & ((e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f))) | (det == 0.0f)) return false;
    p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
    const float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const float lim = t_max * det;
    if (((det < 0.0f) & ((t_scaled >= 0.0f) | (t_scaled < lim))) | ((det > 0.0f) & ((t_scaled <= 0.0f) | (t_scaled > lim)))) return false;
    float inv_det = 1.0f / det;
    float t = t_scaled * inv_det;
    // conservative t > delta_t test, shape.rs:163-185
    float max_z = max_nz(max_nz(fabs_(p0t.z), fabs_(p1t.z)), fabs_(p2t.z)); // (absolute values: no +0 / -0 tie)
    float delta_z = gamma_err(3) * max_z;
    float max_x = max_nz(max_nz(fabs_(p0t.x), fabs_(p1t.x)), fabs_(p2t.x));
    float max_y = max_nz(max_nz(fabs_(p0t.y), fabs_(p1t.y)), fabs_(p2t.y));
    float delta_x = gamma_err(5) * (max_x + max_z);
    float delta_y = gamma_err(5) * (max_y + max_z);
    float delta_e = 2.0f * (gamma_err(2) * max_x * max_y + delta_y * max_x + delta_x * max_y);
    float max_e = max_nz(max_nz(fabs_(e0), fabs_(e1)), fabs_(e2));
    float delta_t = 3.0f * (gamma_err(3) * max_e * max_z + delta_e * max_z + delta_z * max_e) * fabs_(inv_det);
    if (t <= delta_t) return false;
    h.t = t; h.b0 = e0 * inv_det; h.b1 = e1 * inv_det; h.b2 = e2 * inv_det;
    return true;
}
// tri_test_perm in select form: the same arithmetic in the same order, every rejection a term of one predicate instead of an early
// return.  On gfx950 an early return inside a wave's step is a divergent region (exec-mask save / restore, a branch, the waits around
// them: scalar instructions, of which a SIMD issues one per ~4 cycles for all its waves); with a third of a wave's lanes at a leaf
// the whole wave almost never leaves early, so the returns buy nothing there.  What a rejected lane computes past its rejection
// (a division by a zero determinant included) is discarded by the predicate.  The binary64 fallback stays a branch: exact zeros of an
// edge function are rare.  Used by the LDS-form leaf step (ptrs_hip.hip: lf_leaf_step).
PT_HD bool tri_test_perm_sel(f3 p0t, f3 p1t, f3 p2t, float sx, float sy, float sz, float t_max, TriHit &h) {
    p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { // binary64 fallback, shape.rs:124-134
        e0 = (float)((double)p2t.y * (double)p1t.x - (double)p2t.x * (double)p1t.y);
        e1 = (float)((double)p0t.y * (double)p2t.x - (double)p0t.x * (double)p2t.y);
        e2 = (float)((double)p1t.y * (double)p0t.x - (double)p1t.x * (double)p0t.y);
    }
    const float det = e0 + e1 + e2;
    const bool rej_sign = (((e0 < 0.0f) | (e1 < 0.0f) | (e2 < 0.0f)) & ((e0 > 0.0f) | (e1 > 0.0f) | (e2 > 0.0f))) | (det == 0.0f);
    p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
    const float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    const float lim = t_max * det;
    const bool rej_range = ((det < 0.0f) & ((t_scaled >= 0.0f) | (t_scaled < lim))) | ((det > 0.0f) & ((t_scaled <= 0.0f) | (t_scaled > lim)));
    float inv_det = 1.0f / det;
    float t = t_scaled * inv_det;
    float max_z = max_nz(max_nz(fabs_(p0t.z), fabs_(p1t.z)), fabs_(p2t.z));
    float delta_z = gamma_err(3) * max_z;
    float max_x = max_nz(max_nz(fabs_(p0t.x), fabs_(p1t.x)), fabs_(p2t.x));
    float max_y = max_nz(max_nz(fabs_(p0t.y), fabs_(p1t.y)), fabs_(p2t.y));
    float delta_x = gamma_err(5) * (max_x + max_z);
    float delta_y = gamma_err(5) * (max_y + max_z);
    float delta_e = 2.0f * (gamma_err(2) * max_x * max_y + delta_y * max_x + delta_x * max_y);
    float max_e = max_nz(max_nz(fabs_(e0), fabs_(e1)), fabs_(e2));
    float delta_t = 3.0f * (gamma_err(3) * max_e * max_z + delta_e * max_z + delta_z * max_e) * fabs_(inv_det);
    h.t = t; h.b0 = e0 * inv_det; h.b1 = e1 * inv_det; h.b2 = e2 * inv_det;
    return !(rej_sign | rej_range) & !(t <= delta_t);
}
// returns true and fills h when the ray (o, shear of d, t_max) hits triangle (p0,p1,p2) -- shape.rs:85-185
PT_HD bool tri_test_s(f3 o, const RayShear &S, float t_max, f3 p0, f3 p1, f3 p2, TriHit &h) {
    f3 p0t = p0 - o, p1t = p1 - o, p2t = p2 - o;
    const int kz = S.kz;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    p0t = mk3(comp(p0t, kx), comp(p0t, ky), comp(p0t, kz));
    p1t = mk3(comp(p1t, kx), comp(p1t, ky), comp(p1t, kz));
    p2t = mk3(comp(p2t, kx), comp(p2t, ky), comp(p2t, kz));
    return tri_test_perm(p0t, p1t, p2t, S.sx, S.sy, S.sz, t_max, h);
}
// tri_test_s with the test proper in select form (tri_test_perm_sel)
PT_HD bool tri_test_s_sel(f3 o, const RayShear &S, float t_max, f3 p0, f3 p1, f3 p2, TriHit &h) {
    f3 p0t = p0 - o, p1t = p1 - o, p2t = p2 - o;
    const int kz = S.kz;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    p0t = mk3(comp(p0t, kx), comp(p0t, ky), comp(p0t, kz));
    p1t = mk3(comp(p1t, kx), comp(p1t, ky), comp(p1t, kz));
    p2t = mk3(comp(p2t, kx), comp(p2t, ky), comp(p2t, kz));
    return tri_test_perm_sel(p0t, p1t, p2t, S.sx, S.sy, S.sz, t_max, h);
}
// The same with 1/d already at hand (traversal computes it for the slab tests): sz = 1/d[kz] is one of its components --
// the same division, so the same bits -- which saves one of the six IEEE divisions of a ray's set-up.
PT_HD RayShear ray_shear_inv(f3 d, f3 inv) {
    RayShear S;
    S.kz = max_dimension(abs3(d));
    int kx = S.kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    const float dx = comp(d, kx), dy = comp(d, ky), dz = comp(d, S.kz);
    S.sx = -dx / dz; S.sy = -dy / dz; S.sz = comp(inv, S.kz);
    return S;
}
PT_HD bool tri_test(f3 o, f3 d, float t_max, f3 p0, f3 p1, f3 p2, TriHit &h) { return tri_test_s(o, ray_shear(d), t_max, p0, p1, p2, h); }

// triangle partial derivatives, shape.rs:187-215.  Returns false for a degenerate triangle.
PT_HD bool tri_dpduv(f3 p0, f3 p1, f3 p2, f2 uv0, f2 uv1, f2 uv2, f3 &dpdu, f3 &dpdv) {
    f2 duv02 = mk2(uv0.x - uv2.x, uv0.y - uv2.y), duv12 = mk2(uv1.x - uv2.x, uv1.y - uv2.y);
    f3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
    bool degenerate_uv = fabs_(determinant) < 1e-8f;
    dpdu = splat3(0.0f); dpdv = splat3(0.0f);
    if (!degenerate_uv) {
        float invdet = 1.0f / determinant;
        dpdu = (duv12.y * dp02 - duv02.y * dp12) * invdet;
        dpdv = (-duv12.x * dp02 + duv02.x * dp12) * invdet;
    }
    if (degenerate_uv || len2(cross(dpdu, dpdv)) == 0.0f) {
        f3 ng = cross(p2 - p0, p1 - p0);
        if (len2(ng) == 0.0f) return false;
        coordinate_system(normalize(ng), dpdu, dpdv);
    }
    return true;
}

// what the shade stage needs from SurfaceMediumInteraction (interaction.rs:83-101)
struct Surface {
    f3 p, p_error, wo;
    f3 n;           // general.n (geometric normal after face-forwarding to the shading normal)
    f3 ns;          // shading.n
    f3 dpdu, dpdv;  // geometric partials
    f3 s_dpdu, s_dpdv; // shading.dpdu / shading.dpdv
    f2 uv;
    float dudx, dvdx, dudy, dvdy;
    int32_t prim;
    f3 ssn;        // normalize(shading.dpdu) when it is known without computing it (ssn_ok): bsdf.rs:24 takes it as is
    bool ssn_ok;
};

// shape.rs:217-356 + interaction.rs:128-175,193-214 (SurfaceMediumInteraction::new, set_shading_geometry)
PT_HD Surface tri_surface(const TriRegs &T, int32_t prim, float b0, float b1, float b2, f3 wo) {
    Surface s;
    f3 p0 = T.p0, p1 = T.p1, p2 = T.p2;
    f2 uv0 = T.uv0, uv1 = T.uv1, uv2 = T.uv2;
    s.dpdu = T.dpdu; s.dpdv = T.dpdv; // tri_dpduv(p0, p1, p2, uv0, uv1, uv2), evaluated on the host (DTriShade v10-v12)
    float xs = fabs_(b0 * p0.x) + fabs_(b1 * p1.x) + fabs_(b2 * p2.x);
    float ys = fabs_(b0 * p0.y) + fabs_(b1 * p1.y) + fabs_(b2 * p2.y);
    float zs = fabs_(b0 * p0.z) + fabs_(b1 * p1.z) + fabs_(b2 * p2.z);
    s.p_error = gamma_err(7) * mk3(xs, ys, zs);
    s.p = b0 * p0 + b1 * p1 + b2 * p2;
    s.uv = mk2(b0 * uv0.x + b1 * uv1.x + b2 * uv2.x, b0 * uv0.y + b1 * uv1.y + b2 * uv2.y);
    s.wo = wo;
    s.prim = prim;
    s.dudx = s.dvdx = s.dudy = s.dvdy = 0.0f;
    // geometric normal overrides normalize(dpdu x dpdv), shape.rs:260-266: normalize(cross(p0 - p2, p1 - p2)), flipped for
    // reverse_orientation ^ transform_swaps_handedness -- a constant of the triangle (T.ng)
    s.n = T.ng;
    s.ns = s.n;
    s.s_dpdu = s.dpdu; s.s_dpdv = s.dpdv;
    s.ssn = T.ssn; s.ssn_ok = true; // normalize(dpdu)
    if (T.flags & (TRI_HAS_NORMAL | TRI_HAS_TANGENT)) {
        f3 ns;
        if (T.flags & TRI_HAS_NORMAL) {
            ns = b0 * T.n0 + b1 * T.n1 + b2 * T.n2;
            ns = len2(ns) > 0.0f ? normalize(ns) : s.n;
        } else ns = s.n;
        f3 ss;
        if (T.flags & TRI_HAS_TANGENT) {
            ss = b0 * T.s0 + b1 * T.s1 + b2 * T.s2;
            ss = len2(ss) > 0.0f ? normalize(ss) : T.ssn;
        } else ss = T.ssn;
        f3 ts = cross(ss, ns);
        if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, ss, ts);
        if (T.flags & TRI_REVERSE) ts = -ts;
        // set_shading_geometry(ss, ts, .., orientation_is_authoritative = true)
        s.ns = normalize(cross(ss, ts));
        s.n = face_forward(s.n, s.ns);
        s.s_dpdu = ss; s.s_dpdv = ts;
        s.ssn_ok = false;
    }
    return s;
}

// The per-triangle constants of DTriShade v10-v12, computed on the host when the scene is built.
PT_HD void tri_constants(f3 p0, f3 p1, f3 p2, f2 uv0, f2 uv1, f2 uv2, uint32_t flags, f3 &ng, f3 &ssn, f3 &dpdu, f3 &dpdv) {
    tri_dpduv(p0, p1, p2, uv0, uv1, uv2, dpdu, dpdv);
    ng = normalize(cross(p0 - p2, p1 - p2));
    if (((flags & TRI_REVERSE) != 0) != ((flags & TRI_SWAPS) != 0)) ng = -ng;
    ssn = normalize(dpdu);
}

// compute_differentials, interaction.rs:216-281 (only camera rays carry differentials, Q9)
PT_HD void surface_differentials(Surface &s, f3 rx_o, f3 rx_d, f3 ry_o, f3 ry_d) {
    f3 n = s.n, p = s.p;
    float d = dot(n, p);
    float tx = -(dot(n, rx_o) - d) / dot(n, rx_d);
    if (isinf_(tx) || isnan_(tx)) return;
    f3 px = rx_o + tx * rx_d;
    float ty = -(dot(n, ry_o) - d) / dot(n, ry_d);
    if (isinf_(ty) || isnan_(ty)) return;
    f3 py = ry_o + ty * ry_d;
    int d0, d1;
    if (fabs_(n.x) > fabs_(n.y) && fabs_(n.x) > fabs_(n.y)) { d0 = 1; d1 = 2; } // Q10
    else if (fabs_(n.y) > fabs_(n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    float a00 = comp(s.dpdu, d0), a01 = comp(s.dpdv, d0), a10 = comp(s.dpdu, d1), a11 = comp(s.dpdv, d1);
    float bx0 = comp(px, d0) - comp(p, d0), bx1 = comp(px, d1) - comp(p, d1);
    float by0 = comp(py, d0) - comp(p, d0), by1 = comp(py, d1) - comp(p, d1);
    if (!solve_2x2(a00, a01, a10, a11, bx0, bx1, s.dudx, s.dvdx)) { s.dudx = 0.0f; s.dvdx = 0.0f; }
    if (!solve_2x2(a00, a01, a10, a11, by0, by1, s.dudy, s.dvdy)) { s.dudy = 0.0f; s.dvdy = 0.0f; }
}

// Interaction::spawn_ray (interaction.rs:32-39)
PT_HD f3 spawn_origin(f3 p, f3 p_error, f3 n, f3 d) { return offset_ray_origin(p, p_error, n, d); }

} // namespace pt
