// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Parity unpinned.
// Restates src/pathtracer/bsdf.rs, bxdf/{mod,fresnel,microfacet}.rs, sampling.rs:84-126,
// material/{mod,metal,substrate,disney}.rs.
#pragma once
#include "orc_scene.h"

namespace orc {

// bxdf/mod.rs:91-101
enum : uint32_t { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };

// bxdf/mod.rs:11-89
inline float cos_theta(Vec3 w) { return w.z; }
inline float cos_2_theta(Vec3 w) { return w.z * w.z; }
inline float abs_cos_theta(Vec3 w) { return std::fabs(w.z); }
inline float sin_2_theta(Vec3 w) { return fmax_rs(0.0f, 1.0f - cos_2_theta(w)); }
inline float sin_theta(Vec3 w) { return std::sqrt(sin_2_theta(w)); }
inline float tan_2_theta(Vec3 w) { return sin_2_theta(w) / cos_2_theta(w); }
inline float tan_theta(Vec3 w) { return sin_theta(w) / cos_theta(w); }
inline float cos_phi(Vec3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clamp_rs(w.x / s, -1.0f, 1.0f); }
inline float sin_phi(Vec3 w) { float s = sin_theta(w); return s == 0.0f ? 1.0f : clamp_rs(w.y / s, -1.0f, 1.0f); } // sic: 1.0 (mod.rs:51-58)
inline float cos_2_phi(Vec3 w) { return cos_phi(w) * cos_phi(w); }
inline float sin_2_phi(Vec3 w) { return sin_phi(w) * sin_phi(w); }
inline bool same_hemisphere(Vec3 w, Vec3 wp) { return w.z * wp.z > 0.0f; }
inline Vec3 reflect(Vec3 wo, Vec3 n) { return -wo + 2.0f * dot(wo, n) * n; }
inline bool refract(Vec3 wi, Vec3 n, float eta, Vec3 &wt) {
    float cos_theta_i = dot(n, wi);
    float sin_2_theta_i = fmax_rs(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    float sin_2_theta_t = eta * eta * sin_2_theta_i;
    if (sin_2_theta_t > 1.0f) return false;
    float cos_theta_t = std::sqrt(1.0f - sin_2_theta_t);
    wt = eta * -wi + (eta * cos_theta_i - cos_theta_t) * n;
    return true;
}

// sampling.rs:96-122
inline Vec2 concentric_sample_disk(Vec2 u) {
    Vec2 uo{2.0f * u.x - 1.0f, 2.0f * u.y - 1.0f};
    if (uo.x == 0.0f && uo.y == 0.0f) return Vec2{0, 0};
    float theta, r;
    if (std::fabs(uo.x) > std::fabs(uo.y)) { r = uo.x; theta = FRAC_PI_4 * (uo.y / uo.x); }
    else { r = uo.y; theta = FRAC_PI_2 - FRAC_PI_4 * (uo.x / uo.y); }
    return Vec2{r * pt_cosf(theta), r * pt_sinf(theta)};
}
inline Vec3 cosine_sample_hemisphere(Vec2 u) {
    Vec2 d = concentric_sample_disk(u);
    float z = std::sqrt(fmax_rs(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return Vec3(d.x, d.y, z);
}

// bxdf/fresnel.rs:21-64
inline float fr_dielectric(float cos_theta_i, float eta_i, float eta_t) {
    cos_theta_i = clamp_rs(cos_theta_i, -1.0f, 1.0f);
    bool entering = cos_theta_i > 0.0f;
    if (!entering) { std::swap(eta_i, eta_t); cos_theta_i = std::fabs(cos_theta_i); }
    float sin_theta_i = std::sqrt(fmax_rs(0.0f, 1.0f - cos_theta_i * cos_theta_i));
    float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = std::sqrt(fmax_rs(0.0f, 1.0f - sin_theta_t * sin_theta_t));
    float r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
inline Spectrum fr_conductor(float cos_theta_i, Spectrum eta_i, Spectrum eta_t, Spectrum k) {
    cos_theta_i = clamp_rs(cos_theta_i, -1.0f, 1.0f);
    Spectrum eta = eta_t / eta_i, etak = k / eta_i;
    float cos_theta_i2 = cos_theta_i * cos_theta_i;
    float sin_theta_i2 = 1.0f - cos_theta_i2;
    Spectrum eta2 = eta * eta, etak2 = etak * etak;
    Spectrum t0 = eta2 - etak2 - sin_theta_i2;
    Spectrum a2_plus_b2 = sqrt(t0 * t0 + 4.0f * eta2 * etak2);
    Spectrum t1 = a2_plus_b2 + Spectrum(cos_theta_i2); // Spectrum + f32 = add_scalar
    Spectrum a = sqrt(0.5f * (a2_plus_b2 + t0));
    Spectrum t2 = 2.0f * cos_theta_i * a;
    Spectrum rs = (t1 - t2) / (t1 + t2);
    Spectrum t3 = cos_theta_i2 * a2_plus_b2 + Spectrum(sin_theta_i2 * sin_theta_i2);
    Spectrum t4 = t2 * sin_theta_i2;
    Spectrum rp = rs * (t3 - t4) / (t3 + t4);
    return 0.5f * (rp + rs);
}

// material/disney.rs:55-67
inline float schlick_weight(float cos_theta) { float m = clamp_rs(1.0f - cos_theta, 0.0f, 1.0f); return (m * m) * (m * m) * m; }
inline Spectrum fr_schlick_spectrum(Spectrum r0, float cos_theta) { return lerp(r0, Spectrum(1.0f), schlick_weight(cos_theta)); }

enum FresnelKind { FR_NOOP, FR_DIELECTRIC, FR_CONDUCTOR, FR_DISNEY };
struct Fresnel { // fresnel.rs:13-110 + disney.rs:116-136
    FresnelKind kind = FR_NOOP;
    float eta_i = 1, eta_t = 1;   // dielectric
    Spectrum c_eta_i, c_eta_t, k; // conductor
    Spectrum r0; float metallic = 0, eta = 1; // disney
    Spectrum evaluate(float cos_i) const {
        switch (kind) {
            case FR_DIELECTRIC: return Spectrum(fr_dielectric(cos_i, eta_i, eta_t));
            case FR_CONDUCTOR: return fr_conductor(std::fabs(cos_i), c_eta_i, c_eta_t, k);
            case FR_DISNEY: return lerp(Spectrum(fr_dielectric(cos_i, 1.0f, eta)), fr_schlick_spectrum(r0, cos_i), metallic);
            default: return Spectrum(1.0f);
        }
    }
};

// bxdf/microfacet.rs:32-174; disney flag selects DisneyMicrofacetDistribution::g (disney.rs:159-161, Q18)
struct TRDistribution {
    float alpha_x = 0.001f, alpha_y = 0.001f; bool disney = false;
    static TRDistribution make(float ax, float ay, bool disney_) { TRDistribution d; d.alpha_x = fmax_rs(ax, 0.001f); d.alpha_y = fmax_rs(ay, 0.001f); d.disney = disney_; return d; }
    static float roughness_to_alpha(float roughness) {
        roughness = fmax_rs(roughness, 1e-3f);
        float x = pt_logf(roughness);
        return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
    }
    float d(Vec3 wh) const {
        float t2 = tan_2_theta(wh);
        if (std::isinf(t2)) return 0.0f;
        float cos_4_theta = cos_2_theta(wh) * cos_2_theta(wh);
        float e = (cos_2_phi(wh) / (alpha_x * alpha_x) + sin_2_phi(wh) / (alpha_y * alpha_y)) * t2;
        return 1.0f / (PI_F * alpha_x * alpha_y * cos_4_theta * (1.0f + e) * (1.0f + e));
    }
    float lambda(Vec3 w) const {
        float abs_tan_theta = std::fabs(tan_theta(w));
        if (std::isinf(abs_tan_theta)) return 0.0f;
        float alpha = std::sqrt((cos_2_phi(w) * alpha_x * alpha_x) + (sin_2_phi(w) * alpha_y * alpha_y));
        float a2t2 = (alpha * abs_tan_theta) * (alpha * abs_tan_theta);
        return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
    }
    float g1(Vec3 w) const { return 1.0f / (1.0f + lambda(w)); }
    float g(Vec3 wo, Vec3 wi) const { return disney ? g1(wo) * g1(wi) : 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    static void sample_11(float cos_theta_, float u1, float u2, float &slope_x, float &slope_y) {
        if (cos_theta_ > 0.9999f) {
            float r = std::sqrt(u1 / (1.0f - u1));
            float phi = 6.28318530718f * u2;
            slope_x = r * pt_cosf(phi); slope_y = r * pt_sinf(phi);
            return;
        }
        float sin_theta_ = std::sqrt(fmax_rs(0.0f, 1.0f - cos_theta_ * cos_theta_));
        float tan_theta_ = sin_theta_ / cos_theta_;
        float alpha = 1.0f / tan_theta_;
        float g1_ = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (alpha * alpha)));
        float a = 2.0f * u1 / g1_ - 1.0f;
        float tmp = 1.0f / (a * a - 1.0f);
        if (tmp > 1e10f) tmp = 1e10f;
        float b = tan_theta_;
        float d_ = std::sqrt(fmax_rs(0.0f, b * b * tmp * tmp - (a * a - b * b) * tmp));
        float slope_x_1 = b * tmp - d_, slope_x_2 = b * tmp + d_;
        slope_x = (a < 0.0f || slope_x_2 > (1.0f / tan_theta_)) ? slope_x_1 : slope_x_2;
        float s;
        if (u2 > 0.5f) { s = 1.0f; u2 = 2.0f * (u2 - 0.5f); } else { s = -1.0f; u2 = 2.0f * (0.5f - u2); }
        float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) / (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        slope_y = s * z * std::sqrt(1.0f + slope_x * slope_x);
    }
    static Vec3 tr_sample(Vec3 wi, float ax, float ay, float u1, float u2) {
        Vec3 wis = normalize(Vec3(ax * wi.x, ay * wi.y, wi.z));
        float slope_x = 0, slope_y = 0;
        sample_11(cos_theta(wis), u1, u2, slope_x, slope_y);
        float tmp = cos_phi(wis) * slope_x - sin_phi(wis) * slope_y;
        slope_y = sin_phi(wis) * slope_x + cos_phi(wis) * slope_y;
        slope_x = tmp;
        slope_x = ax * slope_x; slope_y = ay * slope_y;
        return normalize(Vec3(-slope_x, -slope_y, 1.0f));
    }
    Vec3 sample_wh(Vec3 wo, Vec2 u) const {
        bool flip = wo.z < 0.0f;
        Vec3 w = flip ? -wo : wo;
        Vec3 wh = tr_sample(w, alpha_x, alpha_y, u.x, u.y);
        return flip ? -wh : wh;
    }
    float pdf(Vec3 wo, Vec3 wh) const { return d(wh) * g1(wo) * std::fabs(dot(wo, wh)) / abs_cos_theta(wo); }
};

enum BxDFKind { BX_LAMBERTIAN, BX_SPECULAR_REFLECTION, BX_SPECULAR_TRANSMISSION, BX_FRESNEL_SPECULAR, BX_MICROFACET_REFLECTION, BX_FRESNEL_BLEND, BX_DISNEY_DIFFUSE };

struct BxDF {
    BxDFKind kind = BX_LAMBERTIAN;
    Spectrum r, t;          // r: Lambertian r / spec R / FresnelSpecular r / MicrofacetReflection r / FresnelBlend rd / DisneyDiffuse r;  t: T / FresnelBlend rs
    float eta_a = 1, eta_b = 1;
    Fresnel fresnel; TRDistribution dist;

    uint32_t type() const {
        switch (kind) {
            case BX_LAMBERTIAN: case BX_DISNEY_DIFFUSE: return BSDF_REFLECTION | BSDF_DIFFUSE;
            case BX_SPECULAR_REFLECTION: return BSDF_REFLECTION | BSDF_SPECULAR;
            case BX_SPECULAR_TRANSMISSION: return BSDF_TRANSMISSION | BSDF_SPECULAR;
            case BX_FRESNEL_SPECULAR: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
            default: return BSDF_REFLECTION | BSDF_GLOSSY;
        }
    }
    bool matches_flags(uint32_t t_) const { return (type() & t_) == type(); }

    Spectrum schlick_fresnel(float cos_theta_) const { // FresnelBlend, microfacet.rs:402-405 (rs stored in t)
        auto pow5 = [](float v) { return (v * v) * (v * v) * v; };
        return t + pow5(1.0f - cos_theta_) * (Spectrum(1.0f) - t);
    }

    Spectrum f(Vec3 wo, Vec3 wi) const {
        switch (kind) {
            case BX_LAMBERTIAN: return r * FRAC_1_PI; // mod.rs:206-208
            case BX_DISNEY_DIFFUSE: { // disney.rs:80-88
                float fo = schlick_weight(abs_cos_theta(wo)), fi = schlick_weight(abs_cos_theta(wi));
                return r * FRAC_1_PI * (1.0f - fo / 2.0f) * (1.0f - fi / 2.0f);
            }
            case BX_MICROFACET_REFLECTION: { // microfacet.rs:197-212
                float cos_theta_o = abs_cos_theta(wo), cos_theta_i = abs_cos_theta(wi);
                Vec3 wh = wi + wo;
                if (cos_theta_i == 0.0f || cos_theta_o == 0.0f) return Spectrum(0.0f);
                if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return Spectrum(0.0f);
                wh = normalize(wh);
                Spectrum F = fresnel.evaluate(dot(wi, wh));
                return r * dist.d(wh) * dist.g(wo, wi) * F / (4.0f * cos_theta_i * cos_theta_o);
            }
            case BX_FRESNEL_BLEND: { // microfacet.rs:408-427
                auto pow5 = [](float v) { return (v * v) * (v * v) * v; };
                Spectrum diffuse = (28.0f / (23.0f * PI_F)) * r * (Spectrum(1.0f) - t) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
                Vec3 wh = wi + wo;
                if (is_zero(wh)) return Spectrum(0.0f);
                wh = normalize(wh);
                Spectrum specular = dist.d(wh) / (4.0f * std::fabs(dot(wi, wh)) * fmax_rs(abs_cos_theta(wi), abs_cos_theta(wo))) * schlick_fresnel(dot(wi, wh));
                return diffuse + specular;
            }
            default: return Spectrum(0.0f); // specular lobes
        }
    }
    float pdf(Vec3 wo, Vec3 wi) const {
        switch (kind) {
            case BX_LAMBERTIAN: case BX_DISNEY_DIFFUSE: // default, mod.rs:173-179
                return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * FRAC_1_PI : 0.0f;
            case BX_MICROFACET_REFLECTION: { // 245-251
                if (!same_hemisphere(wo, wi)) return 0.0f;
                Vec3 wh = normalize(wo + wi);
                return dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
            }
            case BX_FRESNEL_BLEND: { // 460-468
                if (!same_hemisphere(wo, wi)) return 0.0f;
                Vec3 wh = normalize(wo + wi);
                float pdf_wh = dist.pdf(wo, wh);
                return 0.5f * (abs_cos_theta(wi) * FRAC_1_PI + pdf_wh / (4.0f * dot(wo, wh)));
            }
            default: return 0.0f;
        }
    }
    // returns f; pdf stays untouched on the early-outs exactly like the reference (caller zeroes it first)
    Spectrum sample_f(Vec3 wo, Vec3 &wi, Vec2 u, float &pdf_, uint32_t *sampled_type) const {
        switch (kind) {
            case BX_LAMBERTIAN: case BX_DISNEY_DIFFUSE: { // default, mod.rs:106-121
                wi = cosine_sample_hemisphere(u);
                if (wo.z < 0.0f) wi.z *= -1.0f;
                pdf_ = pdf(wo, wi);
                return f(wo, wi);
            }
            case BX_SPECULAR_REFLECTION: { // fresnel.rs:133-144
                wi = Vec3(-wo.x, -wo.y, wo.z); pdf_ = 1.0f;
                return fresnel.evaluate(cos_theta(wi)) * r / abs_cos_theta(wi);
            }
            case BX_SPECULAR_TRANSMISSION: { // fresnel.rs:185-214
                bool entering = cos_theta(wo) > 0.0f;
                float ei = entering ? eta_a : eta_b, et = entering ? eta_b : eta_a;
                if (!refract(wo, face_forward(Vec3(0, 0, 1), wo), ei / et, wi)) return Spectrum(0.0f);
                pdf_ = 1.0f;
                Spectrum ft = t * (Spectrum(1.0f) - Spectrum(fr_dielectric(cos_theta(wi), eta_a, eta_b)));
                ft *= (ei * ei) / (et * et); // TransportMode::Radiance
                return ft / abs_cos_theta(wi);
            }
            case BX_FRESNEL_SPECULAR: { // fresnel.rs:244-288
                float F = fr_dielectric(cos_theta(wo), eta_a, eta_b);
                if (u.x < F) {
                    wi = Vec3(-wo.x, -wo.y, wo.z);
                    if (sampled_type) *sampled_type = BSDF_REFLECTION | BSDF_SPECULAR;
                    pdf_ = F;
                    return F * r / abs_cos_theta(wi);
                }
                bool entering = cos_theta(wo) > 0.0f;
                float ei = entering ? eta_a : eta_b, et = entering ? eta_b : eta_a;
                if (!refract(wo, face_forward(Vec3(0, 0, 1), wo), ei / et, wi)) return Spectrum(0.0f);
                Spectrum ft = t * (Spectrum(1.0f) - F);
                ft *= (ei * ei) / (et * et);
                if (sampled_type) *sampled_type = BSDF_TRANSMISSION | BSDF_SPECULAR;
                pdf_ = 1.0f - F;
                return ft / abs_cos_theta(wi);
            }
            case BX_MICROFACET_REFLECTION: { // microfacet.rs:218-243
                if (wo.z == 0.0f) return Spectrum(0.0f);
                Vec3 wh = dist.sample_wh(wo, u);
                if (dot(wo, wh) < 0.0f) return Spectrum(0.0f);
                wi = reflect(wo, wh);
                if (!same_hemisphere(wo, wi)) return Spectrum(0.0f);
                pdf_ = dist.pdf(wo, wh) / (4.0f * dot(wo, wh));
                return f(wo, wi);
            }
            default: { // BX_FRESNEL_BLEND, microfacet.rs:433-458
                if (u.x < 0.5f) {
                    u.x = fmin_rs(2.0f * u.x, ONE_MINUS_EPSILON);
                    wi = cosine_sample_hemisphere(u);
                    if (wo.z < 0.0f) wi.z *= -1.0f;
                } else {
                    u.x = fmin_rs(2.0f * (u.x - 0.5f), ONE_MINUS_EPSILON);
                    Vec3 wh = dist.sample_wh(wo, u);
                    wi = reflect(wo, wh);
                    if (!same_hemisphere(wo, wi)) return Spectrum(0.0f);
                }
                pdf_ = pdf(wo, wi);
                return f(wo, wi);
            }
        }
    }
};

// bsdf.rs
struct BSDF {
    float eta = 1; Vec3 ns, ng, ss, ts;
    int n_bxdfs = 0; BxDF bxdfs[8];
    static BSDF make(const SurfaceInteraction &si, float eta_) { // 20-34
        BSDF b; b.eta = eta_; b.ns = si.shading.n; b.ss = normalize(si.shading.dpdu); b.ng = si.general.n; b.ts = cross(b.ns, b.ss); return b;
    }
    void add(const BxDF &b) { bxdfs[n_bxdfs++] = b; }
    int num_components(uint32_t flags) const { int n = 0; for (int i = 0; i < n_bxdfs; i++) if (bxdfs[i].matches_flags(flags)) n++; return n; }
    Vec3 world_to_local(Vec3 v) const { return Vec3(dot(v, ss), dot(v, ts), dot(v, ns)); }
    Vec3 local_to_world(Vec3 v) const {
        return Vec3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    // 66-148
    Spectrum sample_f(Vec3 wo_world, Vec3 &wi_world, Vec2 u, float &pdf, uint32_t bxdf_type, uint32_t *sampled_type) const {
        int matching = num_components(bxdf_type);
        if (matching == 0) { pdf = 0.0f; if (sampled_type) *sampled_type = 0; return Spectrum(0.0f); }
        float fc = std::floor(u.x * (float)matching);
        int comp = (int)fc; if (comp > matching - 1) comp = matching - 1;
        const BxDF *bx = nullptr; int count = comp;
        for (int i = 0; i < n_bxdfs; i++) if (bxdfs[i].matches_flags(bxdf_type)) { if (count == 0) { bx = &bxdfs[i]; break; } count--; }
        Vec2 ur{(u.x * (float)matching) - (float)comp, u.y};
        Vec3 wi, wo = world_to_local(wo_world);
        pdf = 0.0f;
        if (sampled_type) *sampled_type = bx->type();
        Spectrum f = bx->sample_f(wo, wi, ur, pdf, sampled_type);
        if (pdf == 0.0f) { if (sampled_type) *sampled_type = 0; return Spectrum(0.0f); }
        wi_world = local_to_world(wi);
        bool spec = (bx->type() & BSDF_SPECULAR) != 0;
        if (!spec && matching > 1)
            for (int i = 0; i < n_bxdfs; i++) if (&bxdfs[i] != bx && bxdfs[i].matches_flags(bxdf_type)) pdf += bxdfs[i].pdf(wo, wi);
        if (matching > 1) pdf /= (float)matching;
        if (!spec && matching > 1) {
            bool refl = dot(wi_world, ng) * dot(wo_world, ng) > 0.0f;
            f = Spectrum(0.0f);
            for (int i = 0; i < n_bxdfs; i++)
                if (bxdfs[i].matches_flags(bxdf_type) && ((refl && (bxdfs[i].type() & BSDF_REFLECTION)) || (!refl && (bxdfs[i].type() & BSDF_TRANSMISSION))))
                    f += bxdfs[i].f(wo, wi);
        }
        return f;
    }
    // 150-187
    Spectrum f(Vec3 wo_w, Vec3 wi_w, uint32_t flags) const {
        Vec3 wi = world_to_local(wi_w), wo = world_to_local(wo_w);
        if (wo.z == 0.0f) return Spectrum(0.0f);
        bool refl = dot(wi_w, ng) * dot(wo_w, ng) > 0.0f;
        Spectrum f_(0.0f);
        for (int i = 0; i < n_bxdfs; i++)
            if (bxdfs[i].matches_flags(flags) && ((refl && (bxdfs[i].type() & BSDF_REFLECTION)) || (!refl && (bxdfs[i].type() & BSDF_TRANSMISSION))))
                f_ += bxdfs[i].f(wo, wi);
        return f_;
    }
    // 189-222
    float pdf(Vec3 wo_world, Vec3 wi_world, uint32_t flags) const {
        if (n_bxdfs == 0) return 0.0f;
        Vec3 wo = world_to_local(wo_world), wi = world_to_local(wi_world);
        if (wo.z == 0.0f) return 0.0f;
        float p = 0.0f; int matching = 0;
        for (int i = 0; i < n_bxdfs; i++) if (bxdfs[i].matches_flags(flags)) { matching++; p += bxdfs[i].pdf(wo, wi); }
        return matching > 0 ? p / (float)matching : 0.0f;
    }
};

inline float sqr(float x) { return x * x; }
inline float schlick_r0_from_eta(float eta) { return sqr(eta - 1.0f) / sqr(eta + 1.0f); } // material/mod.rs:96-98

// material/mod.rs:39-79
inline void normal_mapping(const Texture &d, SurfaceInteraction &si) {
    Vec3 c0 = si.shading.dpdu, c1 = si.shading.dpdv, c2 = si.shading.n;
    Vec3 tn = normalize(d.eval_v3(si));
    // Matrix3 (columns c0,c1,c2) * tn, nalgebra gemv: (c0*x + c1*y) + c2*z per component
    Vec3 v(c0.x * tn.x + c1.x * tn.y + c2.x * tn.z, c0.y * tn.x + c1.y * tn.y + c2.y * tn.z, c0.z * tn.x + c1.z * tn.y + c2.z * tn.z);
    Vec3 ns = normalize(v);
    Vec3 ss = si.shading.dpdu;
    Vec3 ts = cross(ss, ns);
    if (norm_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
    else coordinate_system(ns, ss, ts);
    si.shading.n = ns; si.shading.dpdu = ss; si.shading.dpdv = ts;
}

// Material::compute_scattering_functions; returns false when no BSDF is produced (Q17)
inline bool compute_scattering_functions(const Scene &sc, int32_t mat_id, SurfaceInteraction &si, BSDF &bsdf) {
    const Material &m = sc.materials[mat_id];
    auto T = [&](int k) -> const Texture & { return sc.textures[m.tex[k]]; };
    switch (m.kind) {
        case PTRS_MAT_NORMAL: // mod.rs:136-141
            normal_mapping(T(0), si);
            return compute_scattering_functions(sc, m.inner, si, bsdf);
        case PTRS_MAT_MATTE: { // mod.rs:155-167
            bsdf = BSDF::make(si, 1.0f);
            BxDF b; b.kind = BX_LAMBERTIAN; b.r = T(0).eval_rgb(si); bsdf.add(b);
            return true;
        }
        case PTRS_MAT_MIRROR: { // mod.rs:180-195
            bsdf = BSDF::make(si, 1.0f);
            BxDF b; b.kind = BX_SPECULAR_REFLECTION; b.r = Spectrum(1.0f); b.fresnel.kind = FR_NOOP; bsdf.add(b);
            return true;
        }
        case PTRS_MAT_GLASS: { // mod.rs:216-255
            float eta = T(2).eval_f(si);
            Spectrum r = T(0).eval_rgb(si), t = T(1).eval_rgb(si);
            bsdf = BSDF::make(si, eta);
            if (r.is_black() && t.is_black()) return false;
            BxDF b; b.kind = BX_FRESNEL_SPECULAR; b.r = r; b.t = t; b.eta_a = 1.0f; b.eta_b = eta; bsdf.add(b);
            return true;
        }
        case PTRS_MAT_METAL: { // metal.rs:49-94
            bsdf = BSDF::make(si, 1.0f);
            float u_rough = m.tex[4] >= 0 ? T(4).eval_f(si) : T(3).eval_f(si);
            float v_rough = m.tex[5] >= 0 ? T(5).eval_f(si) : T(3).eval_f(si);
            if (m.flags & 1) { u_rough = TRDistribution::roughness_to_alpha(u_rough); v_rough = TRDistribution::roughness_to_alpha(v_rough); }
            BxDF b; b.kind = BX_MICROFACET_REFLECTION; b.r = T(2).eval_rgb(si);
            b.dist = TRDistribution::make(u_rough, v_rough, false);
            b.fresnel.kind = FR_CONDUCTOR; b.fresnel.c_eta_i = Spectrum(1.0f); b.fresnel.c_eta_t = T(0).eval_rgb(si); b.fresnel.k = T(1).eval_rgb(si);
            bsdf.add(b);
            return true;
        }
        case PTRS_MAT_SUBSTRATE: { // substrate.rs:42-68
            bsdf = BSDF::make(si, 1.0f);
            Spectrum d = T(0).eval_rgb(si), s = T(1).eval_rgb(si);
            float rough_u = T(2).eval_f(si), rough_v = T(3).eval_f(si);
            if (!d.is_black() || s.is_black()) { // Q20
                if (m.flags & 1) { rough_u = TRDistribution::roughness_to_alpha(rough_u); rough_v = TRDistribution::roughness_to_alpha(rough_v); }
                BxDF b; b.kind = BX_FRESNEL_BLEND; b.r = d; b.t = s; b.dist = TRDistribution::make(rough_u, rough_v, false); bsdf.add(b);
            }
            return true;
        }
        default: { // PTRS_MAT_DISNEY, disney.rs:172-264
            bsdf = BSDF::make(si, 1.0f);
            Spectrum c = T(0).eval_rgb(si);
            float metallic_weight = T(1).eval_f(si), e = T(2).eval_f(si);
            float strans = 0.0f;
            float diffuse_weight = (1.0f - metallic_weight) * (1.0f - strans);
            float rough = T(3).eval_f(si);
            float lum = c.y();
            Spectrum c_tint = lum > 0.0f ? c / lum : Spectrum(1.0f);
            if (diffuse_weight > 0.0f) { BxDF b; b.kind = BX_DISNEY_DIFFUSE; b.r = diffuse_weight * c; bsdf.add(b); }
            float aspect = 1.0f;
            float ax = fmax_rs(0.001f, sqr(rough) / aspect), ay = fmax_rs(0.001f, sqr(rough) * aspect);
            float spec_tint = 0.0f;
            Spectrum c_spec_0 = lerp(schlick_r0_from_eta(e) * lerp(Spectrum(1.0f), c_tint, spec_tint), c, metallic_weight);
            BxDF b; b.kind = BX_MICROFACET_REFLECTION; b.r = Spectrum(1.0f); b.dist = TRDistribution::make(ax, ay, true);
            b.fresnel.kind = FR_DISNEY; b.fresnel.r0 = c_spec_0; b.fresnel.metallic = metallic_weight; b.fresnel.eta = e;
            bsdf.add(b);
            return true;
        }
    }
}

} // namespace orc
