// pt_light.h -- the Light trait for the four light kinds: src/pathtracer/light.rs (PointLight
// 86-150, DirectionalLight 152-229, DiffuseAreaLight 231-319, InfiniteAreaLight 321-503),
// Triangle::{sample, pdf_at_point, area} (shape.rs:541-578, 62-72, 533-539),
// Distribution1D/2D (sampling.rs:128-230).
#pragma once
#include "pt_material.h"

namespace pt {

struct LightSample {
    f3 li;       // radiance towards the reference point
    f3 wi;
    float pdf;
    // VisibilityTester end points (light.rs:33-43): p1 of the shadow segment
    f3 p1, p1_err, p1_n;
};

// Triangle::sample (shape.rs:541-578): point, face-forwarded normal, error bound, uv
PT_HD void tri_sample(const TriRegs &T, f2 u, f3 &p, f3 &n, f3 &perr, f2 &uv, const float *n_fixed = nullptr) {
    float su0 = sqrt_(u.x);
    float b0 = 1.0f - su0, b1 = u.y * su0;
    f3 p0 = T.p0, p1 = T.p1, p2 = T.p2;
    float b2 = 1.0f - b0 - b1;
    p = (b0 * p0) + (b1 * p1) + b2 * p2;
    if (n_fixed) n = ld3(n_fixed); // the build checked that the lines below give this for every (b0, b1, b2)
    else {
        n = normalize(cross(p1 - p0, p2 - p0));
        if (T.flags & TRI_HAS_NORMAL) {
            f3 ns = (b0 * T.n0) + (b1 * T.n1) + b2 * T.n2;
            n = face_forward(n, ns);
        } else if (((T.flags & TRI_REVERSE) != 0) != ((T.flags & TRI_SWAPS) != 0)) n = n * -1.0f;
    }
    perr = gamma_err(6) * (abs3(b0 * p0) + abs3(b1 * p1) + abs3(b2 * p2));
    uv = mk2(b0 * T.uv0.x + b1 * T.uv1.x + b2 * T.uv2.x, b0 * T.uv0.y + b1 * T.uv1.y + b2 * T.uv2.y);
}

// the two hit-record fields pdf_at_point needs: p (shape.rs:224) and general.n after
// set_shading_geometry's face-forwarding (shape.rs:261-356, interaction.rs:200-204) -- same
// arithmetic as tri_surface, minus everything that does not feed them
PT_HD void tri_point_normal(const TriRegs &T, float b0, float b1, float b2, f3 &p, f3 &n) {
    f3 p0 = T.p0, p1 = T.p1, p2 = T.p2;
    p = b0 * p0 + b1 * p1 + b2 * p2;
    n = T.ng;
    if (T.flags & (TRI_HAS_NORMAL | TRI_HAS_TANGENT)) {
        f3 ns;
        if (T.flags & TRI_HAS_NORMAL) {
            ns = b0 * T.n0 + b1 * T.n1 + b2 * T.n2;
            ns = len2(ns) > 0.0f ? normalize(ns) : n;
        } else ns = n;
        f3 ss;
        if (T.flags & TRI_HAS_TANGENT) {
            ss = b0 * T.s0 + b1 * T.s1 + b2 * T.s2;
            ss = len2(ss) > 0.0f ? normalize(ss) : T.ssn;
        } else ss = T.ssn;
        f3 ts = cross(ss, ns);
        if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, ss, ts);
        if (T.flags & TRI_REVERSE) ts = -ts;
        n = face_forward(n, normalize(cross(ss, ts)));
    }
}

// Triangle::pdf_at_point (shape.rs:62-72): a single-triangle intersection from the offset origin
template <int FEAT>
PT_HD float tri_pdf_at_point(const DScene &sc, const TriRegs &T, float area, f3 ref_p, const SpawnPair &ref_sp, f3 wi, const float *n_fixed = nullptr) {
    f3 o = spawn_from(ref_sp, wi); // = spawn_origin(ref_p, ref_err, ref_n, wi)
    TriHit h;
    if (!tri_test(o, wi, PT_INF, T.p0, T.p1, T.p2, h)) return 0.0f;
    if (T.flags & TRI_DEGENERATE) return 0.0f;
    if ((FEAT & FEAT_ALPHA) && (T.flags & TRI_HAS_ALPHA)) { // Triangle::intersect's alpha test (shape.rs:227-244)
        const f2 uv = mk2(h.b0 * T.uv0.x + h.b1 * T.uv1.x + h.b2 * T.uv2.x, h.b0 * T.uv0.y + h.b1 * T.uv1.y + h.b2 * T.uv2.y);
        if (tex_eval<FEAT>(sc, T.alpha_tex, uv, 0.0f, 0.0f, 0.0f, 0.0f).x == 0.0f) return 0.0f;
    }
    f3 p, n;
    if (n_fixed) { p = h.b0 * T.p0 + h.b1 * T.p1 + h.b2 * T.p2; n = ld3(n_fixed); }
    else tri_point_normal(T, h.b0, h.b1, h.b2, p, n);
    return len2(ref_p - p) / (fabs_(dot(n, -wi)) * area);
}

PT_HD f3 env_lookup(const DScene &sc, const DLight &L, f2 st) { return tex_lookup_width(sc, sc.texs[L.lmap_tex], st, 0.0f); }
PT_HD float spherical_theta(f3 v) { return pt_acosf(clamp_(v.z, -1.0f, 1.0f)); }
PT_HD float spherical_phi(f3 v) { float p = pt_atan2f(v.y, v.x); return p < 0.0f ? p + 2.0f * PT_PI : p; }

// Distribution1D::sample_continuous (sampling.rs:159-182)
PT_HD float dist1d_sample(const float *func, const float *cdf, float func_int, uint32_t n, float u, float &pdf, uint32_t &off, const float *guide = nullptr, uint32_t g = 0) {
    off = g ? find_interval_cdf_guided(cdf, n + 1, u, guide, g) : find_interval_cdf(cdf, n + 1, u);
    float du = u - cdf[off];
    if ((cdf[off + 1] - cdf[off]) > 0.0f) du /= cdf[off + 1] - cdf[off];
    pdf = func_int > 0.0f ? func[off] / func_int : 0.0f;
    return ((float)off + du) / (float)n;
}

// The environment light's marginal distribution (row integrals, their cdf, the cdf's guide table) from somewhere nearer than the
// global pool: the gfx950 shade kernels keep it in LDS, which takes three links out of the dependent-load chain of a light sample.
struct InfMarginal { const float *func, *cdf, *guide; };

// InfiniteAreaLight::sample_li (light.rs:402-441) up to the point that depends on the shading point: direction, pdf and radiance are
// functions of the two random numbers alone (Distribution2D::sample_continuous, sampling.rs:185-230).  On its own because the gfx950
// pipeline evaluates it for a whole round's vertices in a kernel of its own (k_env_presample), where its chain of ~8 dependent table
// reads runs at full occupancy instead of inside the 2-waves-per-SIMD shade kernels.  false: map_pdf == 0 (light.rs:411-413).
PT_HD bool inf_light_sample(const DScene &sc, const DLight &L, f2 u, f3 &wi, float &pdf, f3 &li, const InfMarginal *im = nullptr) {
    const float *D = sc.distdata;
    float pdf_v, pdf_u; uint32_t v, dummy;
    const float *mf = im ? im->func : D + L.fint_off, *mc = im ? im->cdf : D + L.mcdf_off, *mg = im ? im->guide : D + L.mguide_off; // same tables, same values
    float d1 = dist1d_sample(mf, mc, L.marg_int, (uint32_t)L.nv, u.y, pdf_v, v, mg, L.guide_v);
    float d0 = dist1d_sample(D + L.func_off + (uint64_t)v * (uint32_t)L.nu, D + L.cdf_off + (uint64_t)v * ((uint32_t)L.nu + 1u), mf[v], (uint32_t)L.nu, u.x, pdf_u, dummy,
                             D + L.cguide_off + (uint64_t)v * (L.guide_u + 1u), L.guide_u);
    float map_pdf = pdf_u * pdf_v;
    f3 li_ = splat3(0.0f), wi_ = splat3(0.0f); float pdf_ = 0.0f; // (one exit, the outputs assigned there: see light_sample_li)
    const bool ok = map_pdf != 0.0f;
    if (ok) {
        float theta = d1 * PT_PI, phi = d0 * 2.0f * PT_PI;
        float ct, st, sp, cp; pt_sincosf(theta, &st, &ct); pt_sincosf(phi, &sp, &cp);
        wi_ = xform_vec(L.l2w, mk3(st * cp, st * sp, ct));
        pdf_ = st == 0.0f ? 0.0f : map_pdf / (2.0f * PT_PI * PT_PI * st);
        li_ = env_lookup(sc, L, mk2(d0, d1));
    }
    li = li_; pdf = pdf_; wi = wi_;
    return ok;
}

// Light::sample_li.  Returns false when the reference leaves the visibility tester unset
// (InfiniteAreaLight with map_pdf == 0, light.rs:411-413) -- the reference would panic there.
// INF = false: the caller has every InfiniteAreaLight sample from elsewhere (k_env_presample) and never asks for one here -- the walk
// of the environment map's distribution is then not part of the caller's code (in the 256-register Disney shade kernel it cost 21
// spilled registers and 12 KB of LDS for the marginal table).
template <int FEAT, bool INF = true>
PT_HD bool light_sample_li(const DScene &sc, const DLight &L, f3 ref_p, const SpawnPair &ref_sp, f2 u, LightSample &o, const InfMarginal *im = nullptr) {
    // (one exit, every field of `o` assigned there from locals: with a return per light kind the optimizer merged the kinds' stores of
    // different fields behind a pointer phi, which kept four floats of the sample in scratch memory -- 20 bytes per lane that cost
    // every launch of a shade kernel its scratch set-up)
    f3 li = splat3(0.0f), wi = splat3(0.0f), p1 = ref_p, p1_err = splat3(0.0f), p1_n = splat3(0.0f);
    float pdf = 0.0f;
    bool ok = true;
    if (L.kind == 0) { // point
        f3 pl = ld3(L.v);
        wi = normalize(pl - ref_p); pdf = 1.0f; p1 = pl;
        li = ld3(L.c) / len2(pl - ref_p);
    } else if (L.kind == 1) { // directional
        f3 w = ld3(L.v);
        wi = w; pdf = 1.0f; p1 = ref_p + w * (2.0f * L.world_radius);
        li = ld3(L.c);
    } else if (L.kind == 2) { // diffuse area light on one triangle (its record is embedded in the light)
        const TriRegs T = load_tri_regs(&L.T);
        f3 p, n, perr; f2 uv;
        tri_sample(T, u, p, n, perr, uv, L.n_ok ? L.n_sample : nullptr);
        wi = normalize(p - ref_p);
        pdf = tri_pdf_at_point<FEAT>(sc, T, L.area, ref_p, ref_sp, wi, L.n_ok ? L.n_point : nullptr);
        p1 = p; p1_err = perr; p1_n = n;
        f3 w = -wi;
        li = dot(n, w) > 0.0f ? (L.ke_const ? ld3(L.c) : tex_eval<FEAT>(sc, L.ke_tex, uv, 0.0f, 0.0f, 0.0f, 0.0f)) : splat3(0.0f);
    } else if (!(FEAT & FEAT_INFINITE) || !INF) { ok = false; // unreachable
    } else { // infinite area light
        ok = inf_light_sample(sc, L, u, wi, pdf, li, im);
        p1 = ok ? ref_p + wi * (2.0f * L.world_radius) : ref_p;
    }
    o.li = li; o.wi = wi; o.pdf = pdf; o.p1 = p1; o.p1_err = p1_err; o.p1_n = p1_n;
    return ok;
}

template <int FEAT>
PT_HD float light_pdf_li(const DScene &sc, const DLight &L, f3 ref_p, const SpawnPair &ref_sp, f3 w) {
    if (L.kind == 2) return tri_pdf_at_point<FEAT>(sc, load_tri_regs(&L.T), L.area, ref_p, ref_sp, w, L.n_ok ? L.n_point : nullptr);
    if ((FEAT & FEAT_INFINITE) && L.kind == 3) {
        f3 wi = xform_vec(L.w2l, w);
        float theta = spherical_theta(wi), phi = spherical_phi(wi);
        float st = pt_sinf(theta);
        if (st == 0.0f) return 0.0f;
        // Distribution2D::pdf (sampling.rs:223-229), saturating float->usize casts
        float pu = phi * PT_INV_2PI, pv = theta * PT_INV_PI;
        uint32_t nu = (uint32_t)L.nu, nv = (uint32_t)L.nv;
        float fu = pu * (float)nu, fv = pv * (float)nv;
        uint32_t iu = (fu > 0.0f) ? (fu >= 4294967040.0f ? 0xffffffffu : (uint32_t)fu) : 0u;
        uint32_t iv = (fv > 0.0f) ? (fv >= 4294967040.0f ? 0xffffffffu : (uint32_t)fv) : 0u;
        if (iu > nu - 1u) iu = nu - 1u;
        if (iv > nv - 1u) iv = nv - 1u;
        const float *D = sc.distdata;
        return (D[L.func_off + (uint64_t)iv * nu + iu] / L.marg_int) / (2.0f * PT_PI * PT_PI * st);
    }
    return 0.0f;
}

// Light::le for a ray that escapes (light.rs:45-47, 488-498)
template <int FEAT>
PT_HD f3 light_le(const DScene &sc, const DLight &L, f3 d) {
    if (!(FEAT & FEAT_INFINITE) || L.kind != 3) return splat3(0.0f);
    f3 w = normalize(xform_vec(L.w2l, d));
    return env_lookup(sc, L, mk2(spherical_phi(w) * PT_INV_2PI, spherical_theta(w) * PT_INV_PI));
}

// SurfaceMediumInteraction::le (interaction.rs:297-303) + DiffuseAreaLight::l (light.rs:252-258)
template <int FEAT>
PT_HD f3 surface_le(const DScene &sc, const TriRegs &T, const Surface &s, f3 w) {
    if (T.light < 0) return splat3(0.0f);
    if (dot(s.n, w) > 0.0f) {
        const DLight &L = sc.lights[T.light];
        return L.ke_const ? ld3(L.c) : tex_eval<FEAT>(sc, L.ke_tex, s);
    }
    return splat3(0.0f);
}

PT_HD bool light_is_delta(const DLight &L) { return L.kind == 0 || L.kind == 1; }

} // namespace pt
