// pt_material.h -- Material::compute_scattering_functions for every material kind:
// src/pathtracer/material/mod.rs (normal_mapping 39-79, Matte 143-167, Mirror 169-195, Glass
// 197-256), metal.rs:49-94, substrate.rs:42-68, disney.rs:172-264.
#pragma once
#include "pt_bxdf.h"

namespace pt {

PT_HD float roughness_to_alpha(float roughness) { // microfacet.rs:118-127
    roughness = max_nz(roughness, 1e-3f);
    float x = pt_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
PT_HD void set_tr(Lobe &l, float ax, float ay, bool disney) { l.ax = max_nz(ax, 0.001f); l.ay = max_nz(ay, 0.001f); l.disney_g = disney; }
PT_HD Lobe blank_lobe(int kind) {
    Lobe l; l.kind = kind; l.r = splat3(0.0f); l.t = splat3(0.0f); l.eta_a = 1.0f; l.eta_b = 1.0f; l.fresnel = FR_NOOP;
    l.fa = splat3(0.0f); l.fb = splat3(0.0f); l.ax = 0.001f; l.ay = 0.001f; l.disney_g = false;
    return l;
}

template <int FEAT>
PT_HD void normal_mapping(const DScene &sc, int32_t tex, Surface &s) { // mod.rs:39-79
    f3 c0 = s.s_dpdu, c1 = s.s_dpdv, c2 = s.ns;
    f3 tn = normalize(tex_eval<FEAT>(sc, tex, s));
    f3 v = mk3(c0.x * tn.x + c1.x * tn.y + c2.x * tn.z, c0.y * tn.x + c1.y * tn.y + c2.y * tn.z, c0.z * tn.x + c1.z * tn.y + c2.z * tn.z);
    f3 ns = normalize(v);
    f3 ss = s.s_dpdu;
    f3 ts = cross(ss, ns);
    if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
    else coordinate_system(ns, ss, ts);
    s.ns = ns; s.s_dpdu = ss; s.s_dpdv = ts; s.ssn_ok = false;
}

// texture slot k of material m: the folded constant when the texture is a ConstantTexture
template <int FEAT>
PT_HD f3 mat_tex(const DScene &sc, const DMaterial &m, int k, const Surface &s) {
    if (m.const_mask & (1u << k)) return mk3(m.cval[k][0], m.cval[k][1], m.cval[k][2]);
    return tex_eval<FEAT>(sc, m.tex[k], s);
}

// number of lobe slots per material kind (Disney: DisneyDiffuse + MicrofacetReflection)
template <int MAT> struct MatLobes { static constexpr int N = (MAT == 4) ? 2 : 1; };

// Material::compute_scattering_functions for the (compile-time) kind MAT of the innermost material.
// Returns false when the material yields no BSDF (glass with black r and t, Q17).
template <int MAT, int FEAT>
PT_HD bool make_bsdf(const DScene &sc, int32_t mat_id, Surface &s, BsdfT<MatLobes<MAT>::N> &b) {
    const DMaterial *mp = sc.mats + mat_id;
    if (FEAT & FEAT_NORMAL) { // NormalMaterial wraps another material (mod.rs:136-141)
        for (int guard = 0; guard < 4 && mp->kind == 6; ++guard) {
            normal_mapping<FEAT>(sc, mp->tex[0], s);
            mp = sc.mats + mp->inner;
        }
    }
    const DMaterial &m = *mp;
    if (MAT == 0) { // Matte, mod.rs:155-167
        bsdf_init(b, s, 1.0f);
        Lobe l = blank_lobe(LOBE_LAMBERT); l.r = mat_tex<FEAT>(sc, m, 0, s);
        b.lobe[0] = l; b.n = 1;
        return true;
    } else if (MAT == 2) { // Mirror, mod.rs:180-195
        bsdf_init(b, s, 1.0f);
        Lobe l = blank_lobe(LOBE_SPEC_REFL); l.r = splat3(1.0f);
        b.lobe[0] = l; b.n = 1;
        return true;
    } else if (MAT == 3) { // Glass, mod.rs:216-255
        float eta = mat_tex<FEAT>(sc, m, 2, s).x;
        f3 r = mat_tex<FEAT>(sc, m, 0, s), t = mat_tex<FEAT>(sc, m, 1, s);
        bsdf_init(b, s, eta);
        if (is_black(r) && is_black(t)) return false;
        Lobe l = blank_lobe(LOBE_FRESNEL_SPEC); l.r = r; l.t = t; l.eta_a = 1.0f; l.eta_b = eta;
        b.lobe[0] = l; b.n = 1;
        return true;
    } else if (MAT == 1) { // Metal, metal.rs:49-94
        bsdf_init(b, s, 1.0f);
        float ur = m.tex[4] >= 0 ? mat_tex<FEAT>(sc, m, 4, s).x : mat_tex<FEAT>(sc, m, 3, s).x;
        float vr = m.tex[5] >= 0 ? mat_tex<FEAT>(sc, m, 5, s).x : mat_tex<FEAT>(sc, m, 3, s).x;
        if (m.flags & 1) { ur = roughness_to_alpha(ur); vr = roughness_to_alpha(vr); }
        Lobe l = blank_lobe(LOBE_MICRO_REFL); l.r = mat_tex<FEAT>(sc, m, 2, s);
        set_tr(l, ur, vr, false);
        l.fresnel = FR_CONDUCTOR; l.fa = mat_tex<FEAT>(sc, m, 0, s); l.fb = mat_tex<FEAT>(sc, m, 1, s);
        b.lobe[0] = l; b.n = 1;
        return true;
    } else if (MAT == 5) { // Substrate, substrate.rs:42-68
        bsdf_init(b, s, 1.0f);
        f3 d = mat_tex<FEAT>(sc, m, 0, s), sp = mat_tex<FEAT>(sc, m, 1, s);
        float ru = mat_tex<FEAT>(sc, m, 2, s).x, rv = mat_tex<FEAT>(sc, m, 3, s).x;
        Lobe l = blank_lobe(LOBE_FRESNEL_BLEND);
        if (!is_black(d) || is_black(sp)) { // Q20
            if (m.flags & 1) { ru = roughness_to_alpha(ru); rv = roughness_to_alpha(rv); }
            l.r = d; l.t = sp; set_tr(l, ru, rv, false);
            b.n = 1;
        }
        b.lobe[0] = l;
        return true;
    } else { // Disney, disney.rs:172-264.  Slot 0 = DisneyDiffuse (only when diffuse_weight > 0), then MicrofacetReflection.
        bsdf_init(b, s, 1.0f);
        f3 c = mat_tex<FEAT>(sc, m, 0, s);
        float metallic = mat_tex<FEAT>(sc, m, 1, s).x, e = mat_tex<FEAT>(sc, m, 2, s).x;
        float strans = 0.0f;
        float diffuse_weight = (1.0f - metallic) * (1.0f - strans);
        float rough = mat_tex<FEAT>(sc, m, 3, s).x;
        float lum = luminance(c);
        f3 c_tint = lum > 0.0f ? c / lum : splat3(1.0f);
        float aspect = 1.0f;
        float ax = max_nz(0.001f, (rough * rough) / aspect), ay = max_nz(0.001f, (rough * rough) * aspect);
        float r0s = ((e - 1.0f) * (e - 1.0f)) / ((e + 1.0f) * (e + 1.0f)); // schlick_r0_from_eta
        f3 spec0 = lerp3(r0s * lerp3(splat3(1.0f), c_tint, 0.0f), c, metallic);
        Lobe ld = blank_lobe(LOBE_DISNEY_DIFFUSE); ld.r = diffuse_weight * c;
        Lobe lm = blank_lobe(LOBE_MICRO_REFL); lm.r = splat3(1.0f); set_tr(lm, ax, ay, true);
        lm.fresnel = FR_DISNEY; lm.fa = spec0; lm.fb = mk3(metallic, e, 0.0f);
        // both slots have compile-time lobe kinds; with no diffuse lobe the container must behave as
        // if the microfacet lobe were alone: give slot 0 a type that matches no query instead
        b.lobe[0] = ld; b.lobe[1] = lm; b.n = 2;
        if (!(diffuse_weight > 0.0f)) b.lobe[0].kind = LOBE_ABSENT;
        return true;
    }
}

} // namespace pt
