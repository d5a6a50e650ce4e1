// pt_bxdf.h -- BxDF lobes, Fresnel terms, Trowbridge-Reitz distribution and the BSDF container.
//
// Restates src/pathtracer/bxdf/mod.rs (angle helpers 11-89, default cosine sample_f/pdf
// 106-121,173-179, Lambertian 195-231), bxdf/fresnel.rs (fr_dielectric 21-40, fr_conductor 42-64,
// SpecularReflection 113-150, FresnelSpecular 217-293), bxdf/microfacet.rs (TR 32-174,
// MicrofacetReflection 176-252, FresnelBlend 386-470), material/disney.rs (DisneyDiffuse 73-114,
// DisneyFresnel 116-136, DisneyMicrofacetDistribution 138-170), sampling.rs:96-122, bsdf.rs.
// A BSDF holds at most 2 lobes here: no reference material ever adds more (cap 8, bsdf.rs:7).
#pragma once
#include "pt_texture.h"

namespace pt {

enum : uint32_t { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };
enum : int { LOBE_LAMBERT = 0, LOBE_SPEC_REFL = 1, LOBE_FRESNEL_SPEC = 2, LOBE_MICRO_REFL = 3, LOBE_FRESNEL_BLEND = 4, LOBE_DISNEY_DIFFUSE = 5, LOBE_ABSENT = 6 };
enum : int { FR_NOOP = 0, FR_CONDUCTOR = 1, FR_DISNEY = 2 };

PT_HD float cos2_theta(f3 w) { return w.z * w.z; }
PT_HD float abs_cos(f3 w) { return fabs_(w.z); }
PT_HD float sin2_theta(f3 w) { return max0_(1.0f - cos2_theta(w)); }
PT_HD float sin_theta(f3 w) { return sqrt_(sin2_theta(w)); }
PT_HD float tan2_theta(f3 w) { return sin2_theta(w) / cos2_theta(w); }
PT_HD float tan_theta(f3 w) { return sin_theta(w) / w.z; }
PT_HD float cos_phi(f3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clamp_(w.x / s, -1.0f, 1.0f); }
PT_HD float sin_phi(f3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clamp_(w.y / s, -1.0f, 1.0f); } // sic (mod.rs:51-58)
PT_HD bool same_hemi(f3 a, f3 b) { return a.z * b.z > 0.0f; }
PT_HD f3 reflect_about(f3 wo, f3 n) { return -wo + 2.0f * dot(wo, n) * n; }
PT_HD bool refract_dir(f3 wi, f3 n, float eta, f3 &wt) {
    float ci = dot(n, wi);
    float s2i = max0_(1.0f - ci * ci);
    float s2t = eta * eta * s2i;
    if (s2t > 1.0f) return false;
    float ct = sqrt_(1.0f - s2t);
    wt = eta * -wi + (eta * ci - ct) * n;
    return true;
}
PT_HD f2 concentric_disk(f2 u) {
    float ox = 2.0f * u.x - 1.0f, oy = 2.0f * u.y - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return mk2(0.0f, 0.0f);
    float theta, r;
    if (fabs_(ox) > fabs_(oy)) { r = ox; theta = PT_PI_4 * (oy / ox); }
    else { r = oy; theta = PT_PI_2 - PT_PI_4 * (ox / oy); }
    float sn, cs; pt_sincosf(theta, &sn, &cs);
    return mk2(r * cs, r * sn);
}
PT_HD f3 cosine_hemisphere(f2 u) {
    f2 d = concentric_disk(u);
    return mk3(d.x, d.y, sqrt_(max0_(1.0f - d.x * d.x - d.y * d.y)));
}
PT_HD float fr_dielectric(float ci, float eta_i, float eta_t) {
    ci = clamp_(ci, -1.0f, 1.0f);
    if (!(ci > 0.0f)) { float t = eta_i; eta_i = eta_t; eta_t = t; ci = fabs_(ci); }
    float si = sqrt_(max0_(1.0f - ci * ci));
    float st = eta_i / eta_t * si;
    if (st >= 1.0f) return 1.0f;
    float ct = sqrt_(max0_(1.0f - st * st));
    float r_parl = ((eta_t * ci) - (eta_i * ct)) / ((eta_t * ci) + (eta_i * ct));
    float r_perp = ((eta_i * ci) - (eta_t * ct)) / ((eta_i * ci) + (eta_t * ct));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
PT_HD f3 fr_conductor(float ci, f3 eta_i, f3 eta_t, f3 k) {
    ci = clamp_(ci, -1.0f, 1.0f);
    f3 eta = eta_t / eta_i, etak = k / eta_i;
    float c2 = ci * ci, s2 = 1.0f - c2;
    f3 eta2 = eta * eta, etak2 = etak * etak;
    f3 t0 = add_scalar(eta2 - etak2, -s2);
    f3 a2b2 = sqrt3(t0 * t0 + 4.0f * eta2 * etak2);
    f3 t1 = add_scalar(a2b2, c2);
    f3 a = sqrt3(0.5f * (a2b2 + t0));
    f3 t2 = (2.0f * ci) * a;
    f3 rs = (t1 - t2) / (t1 + t2);
    f3 t3 = add_scalar(c2 * a2b2, s2 * s2);
    f3 t4 = t2 * s2;
    f3 rp = rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (rp + rs);
}
PT_HD float schlick_weight(float c) { float m = clamp_(1.0f - c, 0.0f, 1.0f); return (m * m) * (m * m) * m; }
PT_HD float pow5(float v) { return (v * v) * (v * v) * v; }

struct Lobe {
    int kind;
    f3 r, t;           // see make_* below for the meaning per kind
    float eta_a, eta_b; // FresnelSpecular
    int fresnel;       // FR_* (MicrofacetReflection / SpecularReflection)
    f3 fa, fb;         // conductor: eta_t, k ; disney: r0, (metallic, eta, -)
    float ax, ay;      // TR alphas (already max(.,0.001))
    bool disney_g;     // DisneyMicrofacetDistribution::g (Q18)
};

PT_HD uint32_t lobe_type(const Lobe &l) {
    switch (l.kind) {
        case LOBE_LAMBERT: case LOBE_DISNEY_DIFFUSE: return BSDF_REFLECTION | BSDF_DIFFUSE;
        case LOBE_SPEC_REFL: return BSDF_REFLECTION | BSDF_SPECULAR;
        case LOBE_FRESNEL_SPEC: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
        case LOBE_ABSENT: return 0x100u; // matches no query: an empty slot
        default: return BSDF_REFLECTION | BSDF_GLOSSY;
    }
}
PT_HD bool lobe_matches(const Lobe &l, uint32_t flags) { uint32_t t = lobe_type(l); return (t & flags) == t; }

PT_HD f3 fresnel_eval(const Lobe &l, float ci) {
    if (l.fresnel == FR_CONDUCTOR) return fr_conductor(fabs_(ci), splat3(1.0f), l.fa, l.fb);
    if (l.fresnel == FR_DISNEY) return lerp3(splat3(fr_dielectric(ci, 1.0f, l.fb.y)), lerp3(l.fa, splat3(1.0f), schlick_weight(ci)), l.fb.x);
    return splat3(1.0f);
}
// Trowbridge-Reitz
PT_HD float tr_d(const Lobe &l, f3 wh) {
    float t2 = tan2_theta(wh);
    if (isinf_(t2)) return 0.0f;
    float c4 = cos2_theta(wh) * cos2_theta(wh);
    float cp = cos_phi(wh), sp = sin_phi(wh);
    float e = ((cp * cp) / (l.ax * l.ax) + (sp * sp) / (l.ay * l.ay)) * t2;
    return 1.0f / (PT_PI * l.ax * l.ay * c4 * (1.0f + e) * (1.0f + e));
}
PT_HD float tr_lambda(const Lobe &l, f3 w) {
    float att = fabs_(tan_theta(w));
    if (isinf_(att)) return 0.0f;
    float cp = cos_phi(w), sp = sin_phi(w);
    float alpha = sqrt_(((cp * cp) * l.ax * l.ax) + ((sp * sp) * l.ay * l.ay));
    float a2t2 = (alpha * att) * (alpha * att);
    return (-1.0f + sqrt_(1.0f + a2t2)) / 2.0f;
}
PT_HD float tr_g1(const Lobe &l, f3 w) { return 1.0f / (1.0f + tr_lambda(l, w)); }
PT_HD float tr_g(const Lobe &l, f3 wo, f3 wi) { return l.disney_g ? tr_g1(l, wo) * tr_g1(l, wi) : 1.0f / (1.0f + tr_lambda(l, wo) + tr_lambda(l, wi)); }
PT_HD float tr_pdf(const Lobe &l, f3 wo, f3 wh) { return tr_d(l, wh) * tr_g1(l, wo) * fabs_(dot(wo, wh)) / abs_cos(wo); }
PT_HD void tr_sample11(float ct, float u1, float u2, float &sx, float &sy) {
    if (ct > 0.9999f) {
        float r = sqrt_(u1 / (1.0f - u1));
        float phi = 6.28318530718f * u2;
        float sn, cs; pt_sincosf(phi, &sn, &cs);
        sx = r * cs; sy = r * sn;
        return;
    }
    float st = sqrt_(max0_(1.0f - ct * ct));
    float tt = st / ct;
    float alpha = 1.0f / tt;
    float g1 = 2.0f / (1.0f + sqrt_(1.0f + 1.0f / (alpha * alpha)));
    float a = 2.0f * u1 / g1 - 1.0f;
    float tmp = 1.0f / (a * a - 1.0f);
    if (tmp > 1e10f) tmp = 1e10f;
    float b = tt;
    float dd = sqrt_(max0_(b * b * tmp * tmp - (a * a - b * b) * tmp));
    float s1 = b * tmp - dd, s2 = b * tmp + dd;
    sx = (a < 0.0f || s2 > (1.0f / tt)) ? s1 : s2;
    float s;
    if (u2 > 0.5f) { s = 1.0f; u2 = 2.0f * (u2 - 0.5f); } else { s = -1.0f; u2 = 2.0f * (0.5f - u2); }
    float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    sy = s * z * sqrt_(1.0f + sx * sx);
}
PT_HD f3 tr_sample_wh(const Lobe &l, f3 wo, f2 u) {
    bool flip = wo.z < 0.0f;
    f3 w = flip ? -wo : wo;
    f3 ws = normalize(mk3(l.ax * w.x, l.ay * w.y, w.z));
    float sx = 0.0f, sy = 0.0f;
    tr_sample11(ws.z, u.x, u.y, sx, sy);
    float cp = cos_phi(ws), sp = sin_phi(ws);
    float tmp = cp * sx - sp * sy;
    sy = sp * sx + cp * sy;
    sx = tmp;
    sx = l.ax * sx; sy = l.ay * sy;
    f3 wh = normalize(mk3(-sx, -sy, 1.0f));
    return flip ? -wh : wh;
}

PT_HD f3 lobe_f(const Lobe &l, f3 wo, f3 wi) {
    switch (l.kind) {
        case LOBE_LAMBERT: return l.r * PT_INV_PI;
        case LOBE_DISNEY_DIFFUSE: {
            float fo = schlick_weight(abs_cos(wo)), fi = schlick_weight(abs_cos(wi));
            return l.r * PT_INV_PI * (1.0f - fo / 2.0f) * (1.0f - fi / 2.0f);
        }
        case LOBE_MICRO_REFL: {
            float co = abs_cos(wo), ci = abs_cos(wi);
            f3 wh = wi + wo;
            if (ci == 0.0f || co == 0.0f) return splat3(0.0f);
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return splat3(0.0f);
            wh = normalize(wh);
            f3 F = fresnel_eval(l, dot(wi, wh));
            return l.r * tr_d(l, wh) * tr_g(l, wo, wi) * F / (4.0f * ci * co);
        }
        case LOBE_FRESNEL_BLEND: { // r = rd, t = rs
            f3 diffuse = (28.0f / (23.0f * PT_PI)) * l.r * (splat3(1.0f) - l.t) * (1.0f - pow5(1.0f - 0.5f * abs_cos(wi))) * (1.0f - pow5(1.0f - 0.5f * abs_cos(wo)));
            f3 wh = wi + wo;
            if (is_black(wh)) return splat3(0.0f);
            wh = normalize(wh);
            float c = dot(wi, wh);
            f3 schlick = l.t + pow5(1.0f - c) * (splat3(1.0f) - l.t);
            f3 specular = (tr_d(l, wh) / (4.0f * fabs_(c) * max_nz(abs_cos(wi), abs_cos(wo)))) * schlick;
            return diffuse + specular;
        }
        default: return splat3(0.0f);
    }
}
PT_HD float lobe_pdf(const Lobe &l, f3 wo, f3 wi) {
    switch (l.kind) {
        case LOBE_LAMBERT: case LOBE_DISNEY_DIFFUSE: return same_hemi(wo, wi) ? abs_cos(wi) * PT_INV_PI : 0.0f;
        case LOBE_MICRO_REFL: {
            if (!same_hemi(wo, wi)) return 0.0f;
            f3 wh = normalize(wo + wi);
            return tr_pdf(l, wo, wh) / (4.0f * dot(wo, wh));
        }
        case LOBE_FRESNEL_BLEND: {
            if (!same_hemi(wo, wi)) return 0.0f;
            f3 wh = normalize(wo + wi);
            float pw = tr_pdf(l, wo, wh);
            return 0.5f * (abs_cos(wi) * PT_INV_PI + pw / (4.0f * dot(wo, wh)));
        }
        default: return 0.0f;
    }
}
// pdf is left untouched on early returns (the container zeroes it first, bsdf.rs:100)
PT_HD f3 lobe_sample_f(const Lobe &l, f3 wo, f3 &wi, f2 u, float &pdf, uint32_t &sampled) {
    switch (l.kind) {
        case LOBE_LAMBERT: case LOBE_DISNEY_DIFFUSE: {
            wi = cosine_hemisphere(u);
            if (wo.z < 0.0f) wi.z *= -1.0f;
            pdf = lobe_pdf(l, wo, wi);
            return lobe_f(l, wo, wi);
        }
        case LOBE_SPEC_REFL: {
            wi = mk3(-wo.x, -wo.y, wo.z); pdf = 1.0f;
            return fresnel_eval(l, wi.z) * l.r / abs_cos(wi);
        }
        case LOBE_FRESNEL_SPEC: { // r = R, t = T
            float F = fr_dielectric(wo.z, l.eta_a, l.eta_b);
            if (u.x < F) {
                wi = mk3(-wo.x, -wo.y, wo.z);
                sampled = BSDF_REFLECTION | BSDF_SPECULAR;
                pdf = F;
                return F * l.r / abs_cos(wi);
            }
            bool entering = wo.z > 0.0f;
            float ei = entering ? l.eta_a : l.eta_b, et = entering ? l.eta_b : l.eta_a;
            if (!refract_dir(wo, face_forward(mk3(0.0f, 0.0f, 1.0f), wo), ei / et, wi)) return splat3(0.0f);
            f3 ft = l.t * add_scalar(splat3(1.0f), -F);
            ft = ft * ((ei * ei) / (et * et)); // TransportMode::Radiance
            sampled = BSDF_TRANSMISSION | BSDF_SPECULAR;
            pdf = 1.0f - F;
            return ft / abs_cos(wi);
        }
        case LOBE_MICRO_REFL: {
            if (wo.z == 0.0f) return splat3(0.0f);
            f3 wh = tr_sample_wh(l, wo, u);
            if (dot(wo, wh) < 0.0f) return splat3(0.0f);
            wi = reflect_about(wo, wh);
            if (!same_hemi(wo, wi)) return splat3(0.0f);
            pdf = tr_pdf(l, wo, wh) / (4.0f * dot(wo, wh));
            return lobe_f(l, wo, wi);
        }
        default: { // LOBE_FRESNEL_BLEND
            if (u.x < 0.5f) {
                u.x = min_nz(2.0f * u.x, PT_ONE_MINUS_EPS);
                wi = cosine_hemisphere(u);
                if (wo.z < 0.0f) wi.z *= -1.0f;
            } else {
                u.x = min_nz(2.0f * (u.x - 0.5f), PT_ONE_MINUS_EPS);
                f3 wh = tr_sample_wh(l, wo, u);
                wi = reflect_about(wo, wh);
                if (!same_hemi(wo, wi)) return splat3(0.0f);
            }
            pdf = lobe_pdf(l, wo, wi);
            return lobe_f(l, wo, wi);
        }
    }
}

// ---- BSDF container (bsdf.rs) ---------------------------------------------------------------------
// NL = number of lobe slots of the material (compile time); slot i is live when i < n.  All loops
// are fully unrolled over constant indices so the lobes stay in registers and, with the lobe
// kinds assigned as constants by make_bsdf<MAT>, every switch on `kind` folds away.
template <int NL> struct BsdfT {
    float eta;
    f3 ns, ng, ss, ts;
    int n;
    Lobe lobe[NL];
};
template <int NL> PT_HD void bsdf_init(BsdfT<NL> &b, const Surface &s, float eta) { // bsdf.rs:20-34
    b.eta = eta; b.ns = s.ns; b.ng = s.n; b.ss = s.ssn_ok ? s.ssn : normalize(s.s_dpdu); b.ts = cross(b.ns, b.ss); b.n = 0;
}
template <int NL> PT_HD f3 to_local(const BsdfT<NL> &b, f3 v) { return mk3(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
template <int NL> PT_HD f3 to_world(const BsdfT<NL> &b, f3 v) {
    return mk3(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
template <int NL> PT_HD int bsdf_num(const BsdfT<NL> &b, uint32_t flags) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) c += (i < b.n && lobe_matches(b.lobe[i], flags)) ? 1 : 0;
    return c;
}
PT_HD bool lobe_side_ok(const Lobe &l, bool refl) { uint32_t t = lobe_type(l); return (refl && (t & BSDF_REFLECTION)) || (!refl && (t & BSDF_TRANSMISSION)); }

template <int NL> PT_HD f3 bsdf_f(const BsdfT<NL> &b, f3 wo_w, f3 wi_w, uint32_t flags) { // bsdf.rs:150-187
    f3 wi = to_local(b, wi_w), wo = to_local(b, wo_w);
    if (wo.z == 0.0f) return splat3(0.0f);
    bool refl = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
    f3 f = splat3(0.0f);
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (i < b.n && lobe_matches(b.lobe[i], flags) && lobe_side_ok(b.lobe[i], refl)) f = f + lobe_f(b.lobe[i], wo, wi);
    return f;
}
template <int NL> PT_HD float bsdf_pdf(const BsdfT<NL> &b, f3 wo_w, f3 wi_w, uint32_t flags) { // bsdf.rs:189-222
    if (b.n == 0) return 0.0f;
    f3 wo = to_local(b, wo_w), wi = to_local(b, wi_w);
    if (wo.z == 0.0f) return 0.0f;
    float pdf = 0.0f; int m = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (i < b.n && lobe_matches(b.lobe[i], flags)) { ++m; pdf += lobe_pdf(b.lobe[i], wo, wi); }
    return m > 0 ? pdf / (float)m : 0.0f;
}
// bsdf.rs:66-148.  wi_w is only written when a direction was sampled (like the reference).
template <int NL> PT_HD f3 bsdf_sample_f(const BsdfT<NL> &b, f3 wo_w, f3 &wi_w, f2 u, float &pdf, uint32_t flags, uint32_t &sampled) {
    int m = bsdf_num(b, flags);
    if (m == 0) { pdf = 0.0f; sampled = 0; return splat3(0.0f); }
    int comp_i = (int)floor_(u.x * (float)m);
    if (comp_i > m - 1) comp_i = m - 1;
    int sel = -1, count = comp_i;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (sel < 0 && i < b.n && lobe_matches(b.lobe[i], flags)) { if (count == 0) sel = i; else --count; }
    f2 ur = mk2((u.x * (float)m) - (float)comp_i, u.y);
    f3 wi = splat3(0.0f), wo = to_local(b, wo_w);
    pdf = 0.0f;
    f3 f = splat3(0.0f);
    uint32_t sel_type = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
        if (i == sel) { sel_type = lobe_type(b.lobe[i]); sampled = sel_type; f = lobe_sample_f(b.lobe[i], wo, wi, ur, pdf, sampled); }
    if (pdf == 0.0f) { sampled = 0; return splat3(0.0f); }
    wi_w = to_world(b, wi);
    bool spec = (sel_type & BSDF_SPECULAR) != 0;
    if (!spec && m > 1) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < b.n && i != sel && lobe_matches(b.lobe[i], flags)) pdf += lobe_pdf(b.lobe[i], wo, wi);
    }
    if (m > 1) pdf /= (float)m;
    if (!spec && m > 1) {
        bool refl = dot(wi_w, b.ng) * dot(wo_w, b.ng) > 0.0f;
        f = splat3(0.0f);
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (i < b.n && lobe_matches(b.lobe[i], flags) && lobe_side_ok(b.lobe[i], refl)) f = f + lobe_f(b.lobe[i], wo, wi);
    }
    return f;
}

} // namespace pt
