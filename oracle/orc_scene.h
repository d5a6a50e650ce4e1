// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Parity unpinned.
// Restates, for the flat scene description of include/ptrs.h:
//   src/pathtracer/texture.rs (evaluate paths), shape.rs (Triangle), interaction.rs,
//   primitive.rs, accelerator.rs (SAH build + traversal), light.rs, sampling.rs:84-230.
#pragma once
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "../include/ptrs.h"
#include "orc_math.h"

namespace orc {

struct Counters {
    uint64_t rays_extension = 0, rays_shadow = 0, rays_mis = 0, nodes_visited = 0, tris_tested = 0;
    void add(const Counters &o) {
        rays_extension += o.rays_extension; rays_shadow += o.rays_shadow; rays_mis += o.rays_mis;
        nodes_visited += o.nodes_visited; tris_tested += o.tris_tested;
    }
};

struct Scene;
struct BSDF;

// interaction.rs:8-60
struct Interaction {
    Vec3 p; float time = 0; Vec3 p_error, wo, n;
    Ray spawn_ray(Vec3 d) const { Ray r; r.o = offset_ray_origin(p, p_error, n, d); r.d = d; r.t_max = kInf; return r; }
    Ray spawn_ray_to_it(const Interaction &it2) const {
        Vec3 origin = offset_ray_origin(p, p_error, n, it2.p - p);
        Vec3 target = offset_ray_origin(it2.p, it2.p_error, it2.n, origin - it2.p);
        Ray r; r.o = origin; r.d = target - origin; r.t_max = 1.0f - 0.0001f; return r;
    }
};

struct Shading { Vec3 n, dpdu, dpdv, dndu, dndv; };

// interaction.rs:83-101
struct SurfaceInteraction {
    Interaction general;
    Vec2 uv;
    Vec3 dpdu, dpdv, dndu, dndv;
    Shading shading;
    int32_t prim = -1; // global triangle id (primitive + shape)
    Vec3 dpdx, dpdy;
    float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
    float bary[3] = {0, 0, 0}; // b0,b1,b2 of the accepted hit (shape.rs:157-160); oracle-only debug output

    // interaction.rs:128-175
    static SurfaceInteraction make(Vec3 p, Vec3 p_error, Vec2 uv, Vec3 wo, Vec3 dpdu, Vec3 dpdv, Vec3 dndu, Vec3 dndv, int32_t prim) {
        SurfaceInteraction s;
        Vec3 n = normalize(cross(dpdu, dpdv));
        s.shading.n = n; s.shading.dpdu = dpdu; s.shading.dpdv = dpdv; s.shading.dndu = dndu; s.shading.dndv = dndv;
        s.general.p = p; s.general.p_error = p_error; s.general.wo = wo; s.general.n = n;
        s.uv = uv; s.dpdu = dpdu; s.dpdv = dpdv; s.dndu = dndu; s.dndv = dndv; s.prim = prim;
        return s;
    }
    // interaction.rs:193-214
    void set_shading_geometry(Vec3 dpdus, Vec3 dpdvs, Vec3 dndus, Vec3 dndvs, bool orientation_is_authoritative) {
        shading.n = normalize(cross(dpdus, dpdvs));
        if (orientation_is_authoritative) general.n = face_forward(general.n, shading.n);
        else shading.n = face_forward(shading.n, general.n);
        shading.dpdu = dpdus; shading.dpdv = dpdvs; shading.dndu = dndus; shading.dndv = dndvs;
    }
    // interaction.rs:216-281
    bool compute_differentials(const RayDifferential &ray) {
        if (!ray.has_differentials) return false;
        Vec3 n = general.n, p = general.p;
        float d = dot(n, p);
        float tx = -(dot(n, ray.rx_origin) - d) / dot(n, ray.rx_direction);
        if (std::isinf(tx) || tx != tx) return false;
        Vec3 px = ray.rx_origin + tx * ray.rx_direction;
        float ty = -(dot(n, ray.ry_origin) - d) / dot(n, ray.ry_direction);
        if (std::isinf(ty) || ty != ty) return false;
        Vec3 py = ray.ry_origin + ty * ray.ry_direction;
        dpdx = px - p; dpdy = py - p;
        int dim0, dim1;
        if (std::fabs(n.x) > std::fabs(n.y) && std::fabs(n.x) > std::fabs(n.y)) { dim0 = 1; dim1 = 2; } // Q10: duplicated test
        else if (std::fabs(n.y) > std::fabs(n.z)) { dim0 = 0; dim1 = 2; }
        else { dim0 = 0; dim1 = 1; }
        float a[4] = {dpdu[dim0], dpdv[dim0], dpdu[dim1], dpdv[dim1]};
        float bx[2] = {px[dim0] - p[dim0], px[dim1] - p[dim1]};
        float by[2] = {py[dim0] - p[dim0], py[dim1] - p[dim1]};
        float r[2];
        if (solve_linear_system_2x2(a, bx, r)) { dudx = r[0]; dvdx = r[1]; } else { dudx = 0; dvdx = 0; }
        if (solve_linear_system_2x2(a, by, r)) { dudy = r[0]; dvdy = r[1]; } else { dudy = 0; dvdy = 0; }
        return true;
    }
};

// ---- textures (texture.rs) --------------------------------------------------------------------
struct Texture {
    int kind = PTRS_TEX_CONSTANT, channels = 3;
    float value[3] = {0, 0, 0}, value2[3] = {0, 0, 0};
    float su = 1, sv = 1, du = 0, dv = 0;
    int wrap = PTRS_WRAP_REPEAT;
    std::vector<std::vector<float>> levels; // pyramid
    std::vector<int> cols, rows;

    // texel(): texture.rs:245-273; returns up to 3 channels
    void texel(size_t level, int32_t s, int32_t t, float out[3]) const {
        int nc = cols[level], nr = rows[level];
        out[0] = out[1] = out[2] = 0.0f;
        if (wrap == PTRS_WRAP_REPEAT) { s = abs_mod(s, nc); t = abs_mod(t, nr); }
        else if (wrap == PTRS_WRAP_BLACK) { if (s < 0 || s >= nc || t < 0 || t >= nr) return; }
        else { s = s < 0 ? 0 : (s > nc - 1 ? nc - 1 : s); t = t < 0 ? 0 : (t > nr - 1 ? nr - 1 : t); }
        const float *px = &levels[level][((size_t)t * nc + s) * channels];
        for (int c = 0; c < channels; c++) out[c] = px[c];
    }
    // triangle(): texture.rs:413-428
    void triangle(size_t level, Vec2 st, float out[3]) const {
        if (level > levels.size() - 1) level = levels.size() - 1;
        float s = st.x * (float)cols[level] - 0.5f, t = st.y * (float)rows[level] - 0.5f;
        float s0f = std::floor(s), t0f = std::floor(t);
        float ds = s - s0f, dt = t - t0f;
        int32_t s0 = (int32_t)s0f, t0 = (int32_t)t0f;
        float a[3], b[3], c[3], d[3];
        texel(level, s0, t0, a); texel(level, s0, t0 + 1, b); texel(level, s0 + 1, t0, c); texel(level, s0 + 1, t0 + 1, d);
        for (int k = 0; k < 3; k++)
            out[k] = a[k] * (1.0f - ds) * (1.0f - dt) + b[k] * (1.0f - ds) * dt + c[k] * ds * (1.0f - dt) + d[k] * ds * dt;
    }
    // lookup_width(): texture.rs:447-464
    void lookup_width(Vec2 st, float width, float out[3]) const {
        float level = (float)levels.size() - 1.0f + pt_log2f(fmax_rs(width, 1e-8f));
        if (level < 0.0f) triangle(0, st, out);
        else if (level >= (float)(levels.size() - 1)) triangle(levels.size() - 1, st, out);
        else {
            float il = std::floor(level), delta = level - il;
            size_t i = (size_t)il;
            float a[3], b[3];
            triangle(i, st, a); triangle(i + 1, st, b);
            for (int k = 0; k < 3; k++) out[k] = a[k] * (1.0f - delta) + b[k] * delta; // lerp
        }
    }
    // Texture::evaluate for Constant (15-27), Checker (56-89), Image (185-191 + lookup 430-445)
    void evaluate(const SurfaceInteraction &it, float out[3]) const {
        if (kind == PTRS_TEX_CONSTANT) { out[0] = value[0]; out[1] = value[1]; out[2] = value[2]; return; }
        // UVMap::map 43-53
        Vec2 dst_dx{su * it.dudx, sv * it.dvdx}, dst_dy{su * it.dudy, sv * it.dvdy};
        Vec2 st{su * it.uv.x + du, sv * it.uv.y + dv};
        if (kind == PTRS_TEX_CHECKER) {
            float s_idx = st.x - std::floor(st.x), t_idx = st.y - std::floor(st.y);
            const float *v = ((s_idx <= 0.5f && t_idx <= 0.5f) || (s_idx >= 0.5f && t_idx >= 0.5f)) ? value2 : value;
            out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
            return;
        }
        float width = fmax_rs(fmax_rs(std::fabs(dst_dx.x), std::fabs(dst_dx.y)), fmax_rs(std::fabs(dst_dy.x), std::fabs(dst_dy.y)));
        lookup_width(st, width, out);
    }
    Spectrum eval_rgb(const SurfaceInteraction &it) const { float o[3]; evaluate(it, o); return Spectrum(o[0], o[1], o[2]); }
    float eval_f(const SurfaceInteraction &it) const { float o[3]; evaluate(it, o); return o[0]; }
    Vec3 eval_v3(const SurfaceInteraction &it) const { float o[3]; evaluate(it, o); return Vec3(o[0], o[1], o[2]); }
};

// ---- geometry ---------------------------------------------------------------------------------
struct Mesh {
    std::vector<Vec3> pos, normal, s;
    std::vector<Vec2> uv;
    std::vector<uint32_t> indices;
    int32_t material = -1, alpha_mask = -1;
    bool reverse_orientation = false, transform_swaps_handedness = false;
};

struct Triangle { // shape.rs:7-12 + primitive.rs:20-24
    uint32_t mesh = 0; uint32_t v[3] = {0, 0, 0};
    int32_t area_light = -1; // index into Scene::lights
};

// sampling.rs:128-230
struct Distribution1D {
    std::vector<float> func, cdf; float func_int = 0;
    size_t count() const { return func.size(); }
    float sample_continuous(float u, float &pdf, size_t *off) const {
        size_t offset = find_interval(cdf.size(), [&](size_t i) { return cdf[i] <= u; });
        if (off) *off = offset;
        float du = u - cdf[offset];
        if ((cdf[offset + 1] - cdf[offset]) > 0.0f) du /= cdf[offset + 1] - cdf[offset];
        pdf = func_int > 0.0f ? func[offset] / func_int : 0.0f;
        return ((float)offset + du) / (float)count();
    }
};
struct Distribution2D {
    std::vector<Distribution1D> cond; Distribution1D marginal;
    Vec2 sample_continuous(Vec2 u, float &pdf) const {
        float pdfs[2]; size_t v = 1;
        float d1 = marginal.sample_continuous(u.y, pdfs[1], &v);
        float d0 = cond[v].sample_continuous(u.x, pdfs[0], nullptr);
        pdf = pdfs[0] * pdfs[1];
        return Vec2{d0, d1};
    }
    float pdf(Vec2 p) const {
        size_t nu = cond[0].count(), nv = marginal.count();
        size_t iu = (size_t)(p.x * (float)nu); if (p.x * (float)nu < 0.0f || p.x != p.x) iu = 0; if (iu > nu - 1) iu = nu - 1;
        size_t iv = (size_t)(p.y * (float)nv); if (p.y * (float)nv < 0.0f || p.y != p.y) iv = 0; if (iv > nv - 1) iv = nv - 1;
        return cond[iv].func[iu] / marginal.func_int;
    }
};

struct Light {
    int kind = PTRS_LIGHT_POINT;
    Vec3 v; Spectrum c;
    uint32_t tri = 0; int32_t ke_tex = -1; float area = 0; // AREA
    Vec3 world_center; float world_radius = 0;
    int32_t lmap_tex = -1; float l2w[16], w2l[16]; Distribution2D dist;
    bool is_delta() const { return kind == PTRS_LIGHT_POINT || kind == PTRS_LIGHT_DIRECTIONAL; } // light.rs:29-31
};

struct Material { int kind = PTRS_MAT_MATTE; int32_t tex[6] = {-1, -1, -1, -1, -1, -1}; int32_t flags = 0, inner = -1; };

struct LinearBVHNode { Bounds3 bounds; uint32_t offset = 0; uint16_t num_prims = 0; uint8_t axis = 0; }; // accelerator.rs:89-95

struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Triangle> tris;    // global triangle list, mesh-major = the reference's `primitives` before the build
    std::vector<uint32_t> mesh_first_tri;
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::vector<Light> lights;
    std::vector<uint32_t> infinite_lights;
    std::vector<LinearBVHNode> nodes;
    std::vector<uint32_t> ordered_prims;
    uint32_t bvh_max_depth = 0;

    const Vec3 &P(const Triangle &t, int i) const { return meshes[t.mesh].pos[t.v[i]]; }

    // shape.rs:34-48
    void get_uvs(const Triangle &t, Vec2 uv[3]) const {
        const Mesh &m = meshes[t.mesh];
        if (!m.uv.empty()) { uv[0] = m.uv[t.v[0]]; uv[1] = m.uv[t.v[1]]; uv[2] = m.uv[t.v[2]]; }
        else { uv[0] = Vec2{0, 0}; uv[1] = Vec2{1, 0}; uv[2] = Vec2{1, 1}; }
    }
    Bounds3 tri_bound(const Triangle &t) const { return Bounds3::union_p(Bounds3::from_points(P(t, 0), P(t, 1)), P(t, 2)); } // 526-531
    float tri_area(const Triangle &t) const { return 0.5f * norm(cross(P(t, 1) - P(t, 0), P(t, 2) - P(t, 0))); }          // 533-539

    // Triangle::intersect, shape.rs:74-360.  want_isect=false gives intersect_p (362-524).
    bool tri_intersect(uint32_t prim, const Ray &r, float *t_hit, SurfaceInteraction *isect, bool want_isect, Counters *cnt) const {
        if (cnt) cnt->tris_tested++;
        const Triangle &tr = tris[prim];
        const Mesh &mesh = meshes[tr.mesh];
        Vec3 p0 = P(tr, 0), p1 = P(tr, 1), p2 = P(tr, 2);
        Vec3 p0t = p0 - r.o, p1t = p1 - r.o, p2t = p2 - r.o;
        int kz = max_dimension(vabs(r.d));
        int kx = kz + 1; if (kx == 3) kx = 0;
        int ky = kx + 1; if (ky == 3) ky = 0;
        Vec3 d = permute(r.d, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz); p1t = permute(p1t, kx, ky, kz); p2t = permute(p2t, kx, ky, kz);
        float sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0f / d.z;
        p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
        p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
        p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
        float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { // 124-134
            double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (float)(p1typ0tx - p1txp0ty);
        }
        if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
        float det = e0 + e1 + e2;
        if (det == 0.0f) return false;
        p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
        float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < r.t_max * det)) return false;
        else if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > r.t_max * det)) return false;
        float inv_det = 1.0f / det;
        float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
        float t = t_scaled * inv_det;
        // 163-185
        float max_z_t = fmax_rs(fmax_rs(std::fabs(p0t.z), std::fabs(p1t.z)), std::fabs(p2t.z)); // glm::comp_max
        float delta_z = gamma_n(3) * max_z_t;
        float max_x_t = fmax_rs(fmax_rs(std::fabs(p0t.x), std::fabs(p1t.x)), std::fabs(p2t.x));
        float max_y_t = fmax_rs(fmax_rs(std::fabs(p0t.y), std::fabs(p1t.y)), std::fabs(p2t.y));
        float delta_x = gamma_n(5) * (max_x_t + max_z_t);
        float delta_y = gamma_n(5) * (max_y_t + max_z_t);
        float delta_e = 2.0f * (gamma_n(2) * max_x_t * max_y_t + delta_y * max_x_t + delta_x * max_y_t);
        float max_e = fmax_rs(fmax_rs(std::fabs(e0), std::fabs(e1)), std::fabs(e2));
        float delta_t = 3.0f * (gamma_n(3) * max_e * max_z_t + delta_e * max_z_t + delta_z * max_e) * std::fabs(inv_det);
        if (t <= delta_t) return false;

        if (!want_isect && mesh.alpha_mask < 0) return true; // intersect_p without alpha mask: 470-523

        // 187-215
        Vec3 dpdu, dpdv;
        Vec2 uv[3]; get_uvs(tr, uv);
        Vec2 duv02{uv[0].x - uv[2].x, uv[0].y - uv[2].y}, duv12{uv[1].x - uv[2].x, uv[1].y - uv[2].y};
        Vec3 dp02 = p0 - p2, dp12 = p1 - p2;
        float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = std::fabs(determinant) < 1e-8f;
        if (!degenerate_uv) {
            float invdet = 1.0f / determinant;
            dpdu = (duv12.y * dp02 - duv02.y * dp12) * invdet;
            dpdv = (-duv12.x * dp02 + duv02.x * dp12) * invdet;
        }
        if (degenerate_uv || norm_squared(cross(dpdu, dpdv)) == 0.0f) {
            Vec3 ng = cross(p2 - p0, p1 - p0);
            if (norm_squared(ng) == 0.0f) return false;
            coordinate_system(normalize(ng), dpdu, dpdv);
        }
        // 217-225
        float x_abs_sum = std::fabs(b0 * p0.x) + std::fabs(b1 * p1.x) + std::fabs(b2 * p2.x);
        float y_abs_sum = std::fabs(b0 * p0.y) + std::fabs(b1 * p1.y) + std::fabs(b2 * p2.y);
        float z_abs_sum = std::fabs(b0 * p0.z) + std::fabs(b1 * p1.z) + std::fabs(b2 * p2.z);
        Vec3 p_error = gamma_n(7) * Vec3(x_abs_sum, y_abs_sum, z_abs_sum);
        Vec3 p_hit = b0 * p0 + b1 * p1 + b2 * p2;
        Vec2 uv_hit{b0 * uv[0].x + b1 * uv[1].x + b2 * uv[2].x, b0 * uv[0].y + b1 * uv[1].y + b2 * uv[2].y};
        // 227-244 / 506-520
        if (mesh.alpha_mask >= 0) {
            SurfaceInteraction loc = SurfaceInteraction::make(p_hit, Vec3(), uv_hit, -r.d, dpdu, dpdv, Vec3(), Vec3(), (int32_t)prim);
            if (textures[mesh.alpha_mask].eval_f(loc) == 0.0f) return false;
        }
        if (!want_isect) return true;
        // 246-266
        *isect = SurfaceInteraction::make(p_hit, p_error, uv_hit, -r.d, dpdu, dpdv, Vec3(), Vec3(), (int32_t)prim);
        isect->general.n = normalize(cross(dp02, dp12));
        isect->shading.n = isect->general.n;
        if (mesh.reverse_orientation ^ mesh.transform_swaps_handedness) { isect->general.n = -isect->general.n; isect->shading.n = isect->general.n; }
        // 268-356
        if (!mesh.normal.empty() || !mesh.s.empty()) {
            Vec3 ns;
            if (!mesh.normal.empty()) {
                ns = b0 * mesh.normal[tr.v[0]] + b1 * mesh.normal[tr.v[1]] + b2 * mesh.normal[tr.v[2]];
                if (norm_squared(ns) > 0.0f) ns = normalize(ns); else ns = isect->general.n;
            } else ns = isect->general.n;
            Vec3 ss;
            if (!mesh.s.empty()) {
                ss = b0 * mesh.s[tr.v[0]] + b1 * mesh.s[tr.v[1]] + b2 * mesh.s[tr.v[2]];
                if (norm_squared(ss) > 0.0f) ss = normalize(ss); else ss = normalize(isect->dpdu);
            } else ss = normalize(isect->dpdu);
            Vec3 ts = cross(ss, ns);
            if (norm_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, ss, ts);
            Vec3 dndu, dndv;
            if (!mesh.normal.empty()) {
                Vec3 n0 = mesh.normal[tr.v[0]], n1 = mesh.normal[tr.v[1]], n2 = mesh.normal[tr.v[2]];
                Vec3 dn1 = n0 - n2, dn2 = n1 - n2;
                float det2 = duv02.x * duv12.y - duv02.y * duv12.x;
                if (std::fabs(det2) < 1e-8f) {
                    Vec3 dn = cross(n2 - n0, n1 - n0);
                    if (norm_squared(dn) == 0.0f) { dndu = Vec3(); dndv = Vec3(); }
                    else { Vec3 dnu, dnv; coordinate_system(dn, dnu, dnv); dndu = dnu; dndv = dnv; }
                } else {
                    float id = 1.0f / det2;
                    dndu = (duv12.y * dn1 - duv02.y * dn2) * id;
                    dndv = (-duv12.x * dn1 + duv02.x * dn2) * id;
                }
            }
            if (mesh.reverse_orientation) ts = -ts;
            isect->set_shading_geometry(ss, ts, dndu, dndv, true);
        }
        isect->bary[0] = b0; isect->bary[1] = b1; isect->bary[2] = b2;
        *t_hit = t;
        return true;
    }

    // Triangle::sample, shape.rs:541-578 (+ uniform_sample_triangle 14-17)
    SurfaceInteraction tri_sample(uint32_t prim, Vec2 u) const {
        const Triangle &tr = tris[prim];
        const Mesh &mesh = meshes[tr.mesh];
        float su0 = std::sqrt(u.x);
        float b0 = 1.0f - su0, b1 = u.y * su0;
        Vec3 p0 = P(tr, 0), p1 = P(tr, 1), p2 = P(tr, 2);
        SurfaceInteraction si;
        Interaction &it = si.general;
        it.p = (b0 * p0) + (b1 * p1) + (1.0f - b0 - b1) * p2;
        it.n = normalize(cross(p1 - p0, p2 - p0));
        if (!mesh.normal.empty()) {
            Vec3 ns = (b0 * mesh.normal[tr.v[0]]) + (b1 * mesh.normal[tr.v[1]]) + (1.0f - b0 - b1) * mesh.normal[tr.v[2]];
            it.n = face_forward(it.n, ns);
        } else if (mesh.reverse_orientation ^ mesh.transform_swaps_handedness) it.n = it.n * -1.0f;
        Vec3 p_abs_sum = vabs(b0 * p0) + vabs(b1 * p1) + vabs((1.0f - b0 - b1) * p2);
        it.p_error = gamma_n(6) * p_abs_sum;
        Vec2 uv[3]; get_uvs(tr, uv);
        si.uv = Vec2{b0 * uv[0].x + b1 * uv[1].x + (1.0f - b0 - b1) * uv[2].x, b0 * uv[0].y + b1 * uv[1].y + (1.0f - b0 - b1) * uv[2].y};
        si.prim = (int32_t)prim;
        return si;
    }
    // Triangle::pdf_at_point, shape.rs:62-72 (single-triangle test; not counted as a ray)
    float tri_pdf_at_point(uint32_t prim, const Interaction &ref, Vec3 wi) const {
        Ray ray = ref.spawn_ray(wi);
        float t_hit = 0; SurfaceInteraction isect_light;
        if (!tri_intersect(prim, ray, &t_hit, &isect_light, true, nullptr)) return 0.0f;
        return norm_squared(ref.p - isect_light.general.p) / (std::fabs(dot(isect_light.general.n, -wi)) * tri_area(tris[prim]));
    }

    // ---- BVH (accelerator.rs) -----------------------------------------------------------------
    struct PrimInfo { size_t prim_num; Vec3 centroid; Bounds3 bounds; };
    struct BuildNode { Bounds3 bounds; std::unique_ptr<BuildNode> c[2]; int split_axis = 0; size_t first = 0, n = 0; };

    std::unique_ptr<BuildNode> recursive_build(std::vector<PrimInfo> &info, size_t max_prims, size_t start, size_t end, size_t &total, std::vector<uint32_t> &ordered) {
        total++;
        auto node = std::make_unique<BuildNode>();
        Bounds3 bounds = Bounds3::empty();
        for (size_t i = start; i < end; i++) bounds = Bounds3::union_b(bounds, info[i].bounds);
        size_t n = end - start;
        auto make_leaf = [&]() {
            node->first = ordered.size(); node->n = n; node->bounds = bounds;
            for (size_t i = start; i < end; i++) ordered.push_back((uint32_t)info[i].prim_num);
            return std::move(node);
        };
        if (n == 1) return make_leaf();
        Bounds3 cb = Bounds3::empty();
        for (size_t i = start; i < end; i++) cb = Bounds3::union_p(cb, info[i].centroid);
        int dim = cb.maximum_extent();
        size_t mid;
        if (cb.p_max[dim] == cb.p_min[dim]) return make_leaf();
        if (n <= 2) {
            mid = (start + end) / 2;
            std::nth_element(info.begin() + start, info.begin() + mid, info.begin() + end, [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
        } else {
            constexpr int NB = 12;
            struct Bucket { size_t count = 0; Bounds3 bounds = Bounds3::empty(); } buckets[NB];
            auto bucket_of = [&](const PrimInfo &pi) { size_t b = (size_t)((float)NB * cb.offset(pi.centroid)[dim]); if (b == (size_t)NB) b = NB - 1; return b; };
            for (size_t i = start; i < end; i++) { size_t b = bucket_of(info[i]); buckets[b].count++; buckets[b].bounds = Bounds3::union_b(buckets[b].bounds, info[i].bounds); }
            float cost[NB - 1];
            for (int i = 0; i < NB - 1; i++) {
                Bounds3 b0 = Bounds3::empty(), b1 = Bounds3::empty(); size_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) { b0 = Bounds3::union_b(b0, buckets[j].bounds); c0 += buckets[j].count; }
                for (int j = i + 1; j < NB; j++) { b1 = Bounds3::union_b(b1, buckets[j].bounds); c1 += buckets[j].count; }
                cost[i] = 1.0f + ((float)c0 * b0.surface_area() + (float)c1 * b1.surface_area()) / bounds.surface_area();
            }
            float min_cost = cost[0]; size_t min_b = 0;
            for (int i = 1; i < NB - 1; i++) if (cost[i] < min_cost) { min_cost = cost[i]; min_b = i; }
            float leaf_cost = (float)n;
            if (n > max_prims || min_cost < leaf_cost) {
                auto it = std::partition(info.begin() + start, info.begin() + end, [&](const PrimInfo &pi) { return bucket_of(pi) <= min_b; });
                mid = (size_t)(it - info.begin());
            } else return make_leaf();
        }
        node->split_axis = dim;
        node->c[0] = recursive_build(info, max_prims, start, mid, total, ordered);
        node->c[1] = recursive_build(info, max_prims, mid, end, total, ordered);
        node->bounds = Bounds3::union_b(node->c[0]->bounds, node->c[1]->bounds);
        return node;
    }
    size_t flatten(const BuildNode *n, size_t &offset) {
        size_t my = offset++;
        if (n->n > 0) { nodes[my].bounds = n->bounds; nodes[my].offset = (uint32_t)n->first; nodes[my].num_prims = (uint16_t)n->n; nodes[my].axis = 0; }
        else {
            flatten(n->c[0].get(), offset);
            size_t second = flatten(n->c[1].get(), offset);
            nodes[my].bounds = n->bounds; nodes[my].offset = (uint32_t)second; nodes[my].num_prims = 0; nodes[my].axis = (uint8_t)n->split_axis;
        }
        return my;
    }
    static uint32_t depth_of(const BuildNode *n) { return n ? 1 + std::max(depth_of(n->c[0].get()), depth_of(n->c[1].get())) : 0; }
    void build_bvh(size_t max_prims_in_node = 4) { // BVH::new 103-154 with max_prims_in_node = 4 (mitsuba.rs:361)
        nodes.clear(); ordered_prims.clear();
        if (tris.empty()) return;
        std::vector<PrimInfo> info(tris.size());
        for (size_t i = 0; i < tris.size(); i++) { Bounds3 b = tri_bound(tris[i]); info[i] = PrimInfo{i, b.p_min + 0.5f * (b.p_max - b.p_min), b}; }
        size_t total = 0;
        auto root = recursive_build(info, max_prims_in_node, 0, tris.size(), total, ordered_prims);
        bvh_max_depth = depth_of(root.get());
        nodes.resize(total);
        size_t off = 0;
        flatten(root.get(), off);
    }
    Bounds3 world_bound() const { return nodes.empty() ? Bounds3::empty() : nodes[0].bounds; }

    // BVH::intersect, accelerator.rs:359-417 (+ GeometricPrimitive::intersect primitive.rs:41-51)
    bool intersect(Ray &r, SurfaceInteraction &isect, Counters *cnt) const {
        if (nodes.empty()) return false;
        bool hit = false;
        Vec3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        bool neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
        size_t to_visit = 0, cur = 0, stack[64];
        while (true) {
            const LinearBVHNode &node = nodes[cur];
            if (cnt) cnt->nodes_visited++;
            if (node.bounds.intersect_p_precomp(r, inv_dir, neg)) {
                if (node.num_prims > 0) {
                    for (uint32_t i = 0; i < node.num_prims; i++) {
                        float t_hit = 0;
                        if (tri_intersect(ordered_prims[node.offset + i], r, &t_hit, &isect, true, cnt)) { hit = true; r.t_max = t_hit; }
                    }
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = node.offset; }
                    else { stack[to_visit++] = node.offset; cur = cur + 1; }
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return hit;
    }
    // BVH::intersect_p, accelerator.rs:419-475
    bool intersect_p(const Ray &r, Counters *cnt) const {
        if (nodes.empty()) return false;
        Vec3 inv_dir(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        bool neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
        size_t to_visit = 0, cur = 0, stack[64];
        while (true) {
            const LinearBVHNode &node = nodes[cur];
            if (cnt) cnt->nodes_visited++;
            if (node.bounds.intersect_p_precomp(r, inv_dir, neg)) {
                if (node.num_prims > 0) {
                    for (uint32_t i = 0; i < node.num_prims; i++)
                        if (tri_intersect(ordered_prims[node.offset + i], r, nullptr, nullptr, false, cnt)) return true;
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = node.offset; }
                    else { stack[to_visit++] = node.offset; cur = cur + 1; }
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return false;
    }
    // brute force closest hit: independent cross-check used by tests (not from the reference)
    bool intersect_brute(Ray &r, SurfaceInteraction &isect) const {
        bool hit = false;
        for (uint32_t p = 0; p < tris.size(); p++) { float t = 0; if (tri_intersect(p, r, &t, &isect, true, nullptr)) { hit = true; r.t_max = t; } }
        return hit;
    }

    // SurfaceMediumInteraction::le, interaction.rs:297-303 + DiffuseAreaLight::l light.rs:252-258
    Spectrum le(const SurfaceInteraction &si, Vec3 w) const {
        int32_t al = tris[si.prim].area_light;
        if (al < 0) return Spectrum(0.0f);
        if (dot(si.general.n, w) > 0.0f) return textures[lights[al].ke_tex].eval_rgb(si);
        return Spectrum(0.0f);
    }

    // ---- Light trait (light.rs) ---------------------------------------------------------------
    struct VisibilityTester { Interaction p0, p1; }; // 33-43

    Spectrum light_sample_li(const Light &L, const Interaction &ref, Vec2 u, Vec3 &wi, float &pdf, VisibilityTester &vis) const {
        switch (L.kind) {
            case PTRS_LIGHT_POINT: { // 100-121
                wi = normalize(L.v - ref.p); pdf = 1.0f;
                vis.p0 = ref; vis.p1 = Interaction(); vis.p1.p = L.v; vis.p1.time = ref.time;
                return L.c / norm_squared(L.v - ref.p);
            }
            case PTRS_LIGHT_DIRECTIONAL: { // 175-196
                wi = L.v; pdf = 1.0f;
                Vec3 p_outside = ref.p + L.v * (2.0f * L.world_radius);
                vis.p0 = ref; vis.p1 = Interaction(); vis.p1.p = p_outside; vis.p1.time = ref.time;
                return L.c;
            }
            case PTRS_LIGHT_AREA: { // 261-279
                SurfaceInteraction p_shape = tri_sample(L.tri, u);
                wi = normalize(p_shape.general.p - ref.p);
                pdf = tri_pdf_at_point(L.tri, ref, wi);
                vis.p0 = ref; vis.p1 = p_shape.general;
                Vec3 w = -wi; // DiffuseAreaLight::l(&p_shape, &-wi)
                if (dot(p_shape.general.n, w) > 0.0f) return textures[L.ke_tex].eval_rgb(p_shape);
                return Spectrum(0.0f);
            }
            default: { // INFINITE 402-441
                float map_pdf = 0.0f;
                Vec2 uv = L.dist.sample_continuous(u, map_pdf);
                if (map_pdf == 0.0f) return Spectrum(0.0f); // NOTE: leaves vis unset in the reference (unwrap would panic)
                float theta = uv.y * PI_F, phi = uv.x * 2.0f * PI_F;
                float cos_theta = pt_cosf(theta), sin_theta = pt_sinf(theta);
                float sin_phi = pt_sinf(phi), cos_phi = pt_cosf(phi);
                wi = affine_vector(L.l2w, Vec3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
                if (sin_theta == 0.0f) pdf = 0.0f;
                else pdf = map_pdf / (2.0f * PI_F * PI_F * sin_theta);
                vis.p0 = ref; vis.p1 = Interaction(); vis.p1.p = ref.p + wi * (2.0f * L.world_radius); vis.p1.time = ref.time;
                float o[3]; textures[L.lmap_tex].lookup_width(uv, 0.0f, o);
                return Spectrum(o[0], o[1], o[2]);
            }
        }
    }
    float light_pdf_li(const Light &L, const Interaction &ref, Vec3 w) const {
        switch (L.kind) {
            case PTRS_LIGHT_AREA: return tri_pdf_at_point(L.tri, ref, w); // 285-287
            case PTRS_LIGHT_INFINITE: { // 447-460
                Vec3 wi = affine_vector(L.w2l, w);
                float theta = spherical_theta(wi), phi = spherical_phi(wi);
                float sin_theta = pt_sinf(theta);
                if (sin_theta == 0.0f) return 0.0f;
                return L.dist.pdf(Vec2{phi * INV_2_PI, theta * FRAC_1_PI}) / (2.0f * PI_F * PI_F * sin_theta);
            }
            default: return 0.0f;
        }
    }
    // Light::le for escaping rays: default 0 (45-47); InfiniteAreaLight 488-498
    Spectrum light_le(const Light &L, const Ray &r) const {
        if (L.kind != PTRS_LIGHT_INFINITE) return Spectrum(0.0f);
        Vec3 w = normalize(affine_vector(L.w2l, r.d));
        Vec2 st{spherical_phi(w) * INV_2_PI, spherical_theta(w) * FRAC_1_PI};
        float o[3]; textures[L.lmap_tex].lookup_width(st, 0.0f, o);
        return Spectrum(o[0], o[1], o[2]);
    }
    // VisibilityTester::unoccluded, light.rs:38-42
    bool unoccluded(const VisibilityTester &v, Counters *cnt) const {
        if (cnt) cnt->rays_shadow++;
        return !intersect_p(v.p0.spawn_ray_to_it(v.p1), cnt);
    }

    // ---- construction from the flat description -------------------------------------------------
    int from_desc(const PtrsSceneDesc &d, std::string &err) {
        textures.resize(d.n_textures);
        for (uint32_t i = 0; i < d.n_textures; i++) {
            const PtrsTexture &s = d.textures[i]; Texture &t = textures[i];
            t.kind = s.kind; t.channels = s.channels;
            for (int c = 0; c < 3; c++) { t.value[c] = s.value[c]; t.value2[c] = s.value2[c]; }
            t.su = s.su; t.sv = s.sv; t.du = s.du; t.dv = s.dv; t.wrap = s.wrap;
            if (s.kind == PTRS_TEX_IMAGE) {
                if (s.n_levels <= 0 || !s.level_data) { err = "image texture without levels"; return PTRS_ERR_INVALID; }
                for (int l = 0; l < s.n_levels; l++) {
                    t.cols.push_back(s.level_cols[l]); t.rows.push_back(s.level_rows[l]);
                    size_t n = (size_t)s.level_cols[l] * s.level_rows[l] * s.channels;
                    t.levels.emplace_back(s.level_data[l], s.level_data[l] + n);
                }
            }
        }
        materials.resize(d.n_materials);
        for (uint32_t i = 0; i < d.n_materials; i++) {
            materials[i].kind = d.materials[i].kind; materials[i].flags = d.materials[i].flags; materials[i].inner = d.materials[i].inner;
            for (int k = 0; k < 6; k++) materials[i].tex[k] = d.materials[i].tex[k];
        }
        meshes.resize(d.n_meshes); mesh_first_tri.resize(d.n_meshes);
        for (uint32_t m = 0; m < d.n_meshes; m++) {
            const PtrsMesh &s = d.meshes[m]; Mesh &M = meshes[m];
            M.pos.resize(s.n_verts);
            for (uint32_t v = 0; v < s.n_verts; v++) M.pos[v] = Vec3(s.pos[3 * v], s.pos[3 * v + 1], s.pos[3 * v + 2]);
            if (s.normal) { M.normal.resize(s.n_verts); for (uint32_t v = 0; v < s.n_verts; v++) M.normal[v] = Vec3(s.normal[3 * v], s.normal[3 * v + 1], s.normal[3 * v + 2]); }
            if (s.tangent) { M.s.resize(s.n_verts); for (uint32_t v = 0; v < s.n_verts; v++) M.s[v] = Vec3(s.tangent[3 * v], s.tangent[3 * v + 1], s.tangent[3 * v + 2]); }
            if (s.uv) { M.uv.resize(s.n_verts); for (uint32_t v = 0; v < s.n_verts; v++) M.uv[v] = Vec2{s.uv[2 * v], s.uv[2 * v + 1]}; }
            M.indices.assign(s.indices, s.indices + 3 * (size_t)s.n_tris);
            M.material = s.material; M.alpha_mask = s.alpha_mask_tex;
            M.reverse_orientation = s.reverse_orientation != 0; M.transform_swaps_handedness = s.transform_swaps_handedness != 0;
            mesh_first_tri[m] = (uint32_t)tris.size();
            for (uint32_t t = 0; t < s.n_tris; t++) {
                Triangle T; T.mesh = m; T.v[0] = s.indices[3 * t]; T.v[1] = s.indices[3 * t + 1]; T.v[2] = s.indices[3 * t + 2];
                for (int k = 0; k < 3; k++) if (T.v[k] >= s.n_verts) { err = "vertex index out of range"; return PTRS_ERR_INVALID; }
                tris.push_back(T);
            }
        }
        lights.resize(d.n_lights);
        for (uint32_t i = 0; i < d.n_lights; i++) {
            const PtrsLight &s = d.lights[i]; Light &L = lights[i];
            L.kind = s.kind; L.v = Vec3(s.v[0], s.v[1], s.v[2]); L.c = Spectrum(s.c[0], s.c[1], s.c[2]);
            L.world_center = Vec3(s.world_center[0], s.world_center[1], s.world_center[2]); L.world_radius = s.world_radius;
            if (s.kind == PTRS_LIGHT_AREA) {
                if (s.mesh >= d.n_meshes || s.tri >= d.meshes[s.mesh].n_tris) { err = "area light triangle out of range"; return PTRS_ERR_INVALID; }
                L.tri = mesh_first_tri[s.mesh] + s.tri; L.ke_tex = s.ke_tex; L.area = tri_area(tris[L.tri]);
                tris[L.tri].area_light = (int32_t)i;
            } else if (s.kind == PTRS_LIGHT_INFINITE) {
                L.lmap_tex = s.lmap_tex;
                std::memcpy(L.l2w, s.light_to_world, 64); std::memcpy(L.w2l, s.world_to_light, 64);
                size_t nu = s.dist_nu, nv = s.dist_nv;
                L.dist.cond.resize(nv);
                for (size_t v = 0; v < nv; v++) {
                    L.dist.cond[v].func.assign(s.dist_func + v * nu, s.dist_func + (v + 1) * nu);
                    L.dist.cond[v].cdf.assign(s.dist_cdf + v * (nu + 1), s.dist_cdf + (v + 1) * (nu + 1));
                    L.dist.cond[v].func_int = s.dist_func_int[v];
                }
                L.dist.marginal.func.assign(s.dist_func_int, s.dist_func_int + nv);
                L.dist.marginal.cdf.assign(s.marg_cdf, s.marg_cdf + nv + 1);
                L.dist.marginal.func_int = s.marg_func_int;
                infinite_lights.push_back(i);
            }
        }
        if (d.bvh_nodes && d.n_bvh_nodes) {
            nodes.resize(d.n_bvh_nodes);
            for (uint32_t i = 0; i < d.n_bvh_nodes; i++) {
                const PtrsBvhNode &s = d.bvh_nodes[i];
                nodes[i].bounds.p_min = Vec3(s.p_min[0], s.p_min[1], s.p_min[2]); nodes[i].bounds.p_max = Vec3(s.p_max[0], s.p_max[1], s.p_max[2]);
                nodes[i].offset = s.offset; nodes[i].num_prims = s.num_prims; nodes[i].axis = s.axis;
            }
            ordered_prims.assign(d.bvh_prims, d.bvh_prims + tris.size());
        } else build_bvh(4);
        return PTRS_OK;
    }
};

} // namespace orc
