// ptrs_host.cpp -- see ptrs_host.hpp.  Plain C++17; binary32 arithmetic in the order nalgebra 0.32.2
// performs it (same order as pathtracer-rs_amd/scene.py, so both hosts emit identical bits).
#include "ptrs_host.hpp"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include "../../include/ptrs_detmath.h"

namespace ptrs_host {

// ---- Film ------------------------------------------------------------------------------------------
void Film::clear() { for (auto &p : pixels) p = PtrsFilmPixel{{0, 0, 0}, 0}; }
void Film::get_sample_bounds(int32_t out[4]) const {
    const float r = 2.0f;
    out[0] = (int32_t)std::floor(0.5f - r); out[1] = (int32_t)std::floor(0.5f - r);
    out[2] = (int32_t)std::ceil((float)width - 0.5f + r); out[3] = (int32_t)std::ceil((float)height - 0.5f + r);
}
static float gamma_correct(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * pt_powf(v, 1.0f / 2.4f) - 0.055f; } // math.rs:133-139
static uint8_t to_u8(float v) { // spectrum.rs:95-102: (g*255 + 0.5).clamp(0,255) as u8
    float x = gamma_correct(v) * 255.0f + 0.5f;
    if (!(x == x)) return 0; // NaN as u8 saturates to 0 in Rust
    if (x < 0.0f) x = 0.0f;
    if (x > 255.0f) x = 255.0f;
    return (uint8_t)x;
}
std::vector<uint8_t> Film::to_rgba_image() const {
    std::vector<uint8_t> out((size_t)width * height * 4);
    for (size_t i = 0; i < pixels.size(); ++i) {
        const float inv_wt = 1.0f / pixels[i].weight;
        out[4 * i] = to_u8(pixels[i].rgb[0] * inv_wt); out[4 * i + 1] = to_u8(pixels[i].rgb[1] * inv_wt);
        out[4 * i + 2] = to_u8(pixels[i].rgb[2] * inv_wt); out[4 * i + 3] = 255;
    }
    return out;
}

// ---- scene description --------------------------------------------------------------------------------
size_t RenderScene::num_triangles() const { size_t n = 0; for (auto &m : meshes) n += m.indices.size() / 3; return n; }
const PtrsSceneDesc &RenderScene::desc() {
    // Light::preprocess (light.rs:209-211,480-482) with Bounds3::bounding_sphere (bounds.rs:126-134)
    float lo[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f}, hi[3] = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    for (auto &m : meshes) for (size_t i = 0; i < m.pos.size(); ++i) { int c = (int)(i % 3); lo[c] = std::fmin(lo[c], m.pos[i]); hi[c] = std::fmax(hi[c], m.pos[i]); }
    float center[3], d[3];
    for (int c = 0; c < 3; ++c) { center[c] = (lo[c] + hi[c]) * 0.5f; d[c] = center[c] - hi[c]; }
    const float radius = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    for (auto &l : lights) if (l.kind == PTRS_LIGHT_DIRECTIONAL || l.kind == PTRS_LIGHT_INFINITE) { std::memcpy(l.world_center, center, 12); l.world_radius = radius; }
    abi_meshes_.clear();
    for (auto &m : meshes) {
        PtrsMesh a{};
        a.n_verts = (uint32_t)(m.pos.size() / 3); a.n_tris = (uint32_t)(m.indices.size() / 3);
        a.pos = m.pos.data(); a.normal = m.normal.empty() ? nullptr : m.normal.data(); a.tangent = m.tangent.empty() ? nullptr : m.tangent.data();
        a.uv = m.uv.empty() ? nullptr : m.uv.data(); a.indices = m.indices.data();
        a.material = m.material; a.alpha_mask_tex = m.alpha_mask_tex;
        abi_meshes_.push_back(a);
    }
    desc_ = PtrsSceneDesc{};
    desc_.n_meshes = (uint32_t)abi_meshes_.size(); desc_.meshes = abi_meshes_.data();
    desc_.n_materials = (uint32_t)materials.size(); desc_.materials = materials.data();
    desc_.n_textures = (uint32_t)textures.size(); desc_.textures = textures.data();
    desc_.n_lights = (uint32_t)lights.size(); desc_.lights = lights.data();
    return desc_;
}

// ---- sampler / integrator -----------------------------------------------------------------------------
static int round_up_pow2(int v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
SamplerBuilder::SamplerBuilder(int spp, const int32_t bounds[4]) {
    samples_per_pixel = round_up_pow2(spp < 1 ? 1 : spp);
    if (samples_per_pixel != spp) std::fprintf(stderr, "WARN non power-of-two sample count rounded up to %d for sobol sampler\n", samples_per_pixel);
    std::memcpy(sample_bounds, bounds, 16);
    int ext = std::max(bounds[2] - bounds[0], bounds[3] - bounds[1]);
    resolution = round_up_pow2(ext);
    log_2_resolution = 31u - (uint32_t)__builtin_clz((uint32_t)resolution);
}
PathIntegrator::PathIntegrator(const SamplerBuilder &sb, int max_depth, bool show_progress_bar, int device)
    : sb_(sb), max_depth_(max_depth), show_progress_bar_(show_progress_bar), device_(device) {}
PathIntegrator::~PathIntegrator() {
    if (gpu_scene_) ptrs_scene_destroy(gpu_scene_);
    for (PtrsScene *s : multi_scenes_) if (s) ptrs_scene_destroy(s);
}
void PathIntegrator::preprocess(const RenderScene &scene) {
    if (scene.lights.size() > 16) std::fprintf(stderr, "WARN scene contains too many lights for path integrator to handle well\n");
}
PtrsRenderParams PathIntegrator::params(const Camera &camera) const {
    PtrsRenderParams p{};
    p.width = camera.film.width; p.height = camera.film.height; p.spp = sb_.samples_per_pixel; p.max_depth = max_depth_;
    p.rr_threshold = rr_threshold_; p.rr_start_depth = rr_start_depth_; p.rr_enable = rr_enable_ ? 1 : 0;
    p.row_begin = 0; p.row_end = camera.film.height; p.device = device_; p.paths_per_pass = 0; p.flags = 0;
    return p;
}
int PathIntegrator::render(Camera &camera, RenderScene &scene, PtrsStats *stats) {
    if (!gpu_scene_ || gpu_scene_src_ != &scene) {
        if (gpu_scene_) { ptrs_scene_destroy(gpu_scene_); gpu_scene_ = nullptr; }
        int rc = ptrs_scene_create(&scene.desc(), device_, &gpu_scene_);
        if (rc != PTRS_OK) { last_error = ptrs_last_error(); return rc; }
        gpu_scene_src_ = &scene;
    }
    const PtrsRenderParams p = params(camera);
    int rc;
    if (on_pass || show_progress_bar_) { // integrator.rs:631-634 (progress bar) / headless.rs:197-214 (film read while rendering): the film is published pass by pass
        struct Ctx { PathIntegrator *self; } ctx{this};
        auto thunk = [](void *user, uint32_t done, uint32_t total, int32_t y0, int32_t y1) {
            PathIntegrator *self = static_cast<Ctx *>(user)->self;
            if (self->show_progress_bar_) std::fprintf(stderr, "\rrendering: pass %u/%u%s", done, total, done == total ? "\n" : "");
            if (self->on_pass) self->on_pass(done, total, y0, y1);
        };
        rc = ptrs_render_progressive(gpu_scene_, &camera.abi, &p, camera.film.pixels.data(), thunk, &ctx, stats);
    } else rc = ptrs_render(gpu_scene_, &camera.abi, &p, camera.film.pixels.data(), stats);
    if (rc != PTRS_OK) last_error = ptrs_last_error();
    return rc;
}

bool probe_row_cost(PtrsScene *scene, const PtrsCamera &camera, const PtrsRenderParams &params, int /* strips: the old form's; ignored */, std::vector<float> &row_cost, std::string &err) {
    row_cost.assign((size_t)params.height, 0.0f);
    PtrsRenderParams p = params;
    p.spp = 1;
    PtrsStats st{};
    if (ptrs_render_row_cost(scene, &camera, &p, row_cost.data(), &st) != PTRS_OK) { err = ptrs_last_error(); return false; } // one 1-spp render with a device counter per sample row
    return true;
}

int PathIntegrator::render_multi(Camera &camera, RenderScene &scene, int n_devices, bool cost_weighted, PtrsStats *stats, std::vector<int32_t> *bands_out) {
    if (n_devices < 1) { last_error = "render_multi needs at least one device"; return PTRS_ERR_INVALID; }
    if (multi_src_ != &scene || (int)multi_scenes_.size() != n_devices) {
        for (PtrsScene *s : multi_scenes_) if (s) ptrs_scene_destroy(s);
        multi_scenes_.assign((size_t)n_devices, nullptr);
        for (int d = 0; d < n_devices; ++d) { // the scene is replicated: every device walks the whole BVH for its band
            int rc = ptrs_scene_create(&scene.desc(), d, &multi_scenes_[(size_t)d]);
            if (rc != PTRS_OK) { last_error = std::string("device ") + std::to_string(d) + ": " + ptrs_last_error(); return rc; }
        }
        multi_src_ = &scene; plan_key_.clear();
    }
    const PtrsRenderParams p = params(camera);
    std::vector<int32_t> bounds((size_t)n_devices + 1);
    std::vector<float> cost;
    int rc = PTRS_OK;
    if (cost_weighted && n_devices > 1) {
        // the plan of a view is made once: a second frame of the same scene, camera, resolution and depth on the same devices pays nothing
        std::string key((const char *)&camera.abi, sizeof(camera.abi));
        key.append((const char *)&p.width, sizeof(p.width)).append((const char *)&p.height, sizeof(p.height)).append((const char *)&p.max_depth, sizeof(p.max_depth)).append((const char *)&n_devices, sizeof(n_devices));
        if (key != plan_key_) {
            if (!probe_row_cost(multi_scenes_[0], camera.abi, p, 0, cost, last_error)) return PTRS_ERR_DEVICE;
            std::vector<int32_t> planned((size_t)n_devices + 1), even((size_t)n_devices + 1);
            if ((rc = ptrs_plan_bands(p.height, (uint32_t)n_devices, cost.data(), planned.data())) != PTRS_OK || (rc = ptrs_plan_bands(p.height, (uint32_t)n_devices, nullptr, even.data())) != PTRS_OK) { last_error = ptrs_last_error(); return rc; }
            auto worst = [&](const std::vector<int32_t> &b) { double w = 0.0; for (int d = 0; d < n_devices; ++d) { double c = 0.0; for (int32_t y = b[(size_t)d]; y < b[(size_t)d + 1]; ++y) c += cost[(size_t)y]; w = std::max(w, c); } return w; };
            plan_ = worst(even) > 1.01 * worst(planned) ? planned : even; // equal bands unless the plan takes at least 1 % off the slowest band
            plan_key_ = key;
        }
        bounds = plan_;
    } else if ((rc = ptrs_plan_bands(p.height, (uint32_t)n_devices, nullptr, bounds.data())) != PTRS_OK) { last_error = ptrs_last_error(); return rc; }
    if (bands_out) *bands_out = bounds;
    rc = ptrs_render_multi(multi_scenes_.data(), (uint32_t)n_devices, &camera.abi, &p, bounds.data(), camera.film.pixels.data(), stats);
    if (rc != PTRS_OK) last_error = ptrs_last_error();
    return rc;
}

// ---- minimal XML ------------------------------------------------------------------------------------------
struct Node {
    std::string name;
    std::map<std::string, std::string> attr;
    std::vector<std::unique_ptr<Node>> kids;
    const Node *child(const std::string &n) const { for (auto &k : kids) if (k->name == n) return k.get(); return nullptr; }
    const Node *child_named(const std::string &n, const std::string &name_attr) const {
        for (auto &k : kids) if (k->name == n) { auto it = k->attr.find("name"); if (it != k->attr.end() && it->second == name_attr) return k.get(); }
        return nullptr;
    }
    std::string get(const std::string &a, const std::string &def = "") const { auto it = attr.find(a); return it == attr.end() ? def : it->second; }
};
struct Parser {
    const std::string &s; size_t i = 0; std::string err;
    explicit Parser(const std::string &src) : s(src) {}
    void ws() { while (i < s.size() && std::isspace((unsigned char)s[i])) ++i; }
    bool skip_misc() { // comments, declarations, text
        for (;;) {
            while (i < s.size() && s[i] != '<') ++i;
            if (i >= s.size()) return false;
            if (s.compare(i, 4, "<!--") == 0) { size_t e = s.find("-->", i); if (e == std::string::npos) return false; i = e + 3; continue; }
            if (s.compare(i, 2, "<?") == 0) { size_t e = s.find("?>", i); if (e == std::string::npos) return false; i = e + 2; continue; }
            return true;
        }
    }
    std::unique_ptr<Node> element() {
        if (!skip_misc() || s[i] != '<' || s[i + 1] == '/') return nullptr;
        ++i;
        auto n = std::make_unique<Node>();
        while (i < s.size() && !std::isspace((unsigned char)s[i]) && s[i] != '>' && s[i] != '/') n->name += s[i++];
        for (;;) {
            ws();
            if (i >= s.size()) { err = "unexpected end of file"; return nullptr; }
            if (s[i] == '/') { i += 2; return n; } // "/>"
            if (s[i] == '>') { ++i; break; }
            std::string key;
            while (i < s.size() && s[i] != '=' && !std::isspace((unsigned char)s[i])) key += s[i++];
            ws(); if (s[i] != '=') { err = "attribute without value"; return nullptr; }
            ++i; ws();
            char q = s[i++];
            std::string val;
            while (i < s.size() && s[i] != q) val += s[i++];
            ++i;
            n->attr[key] = val;
        }
        for (;;) {
            if (!skip_misc()) { err = "unterminated element " + n->name; return nullptr; }
            if (s[i + 1] == '/') { size_t e = s.find('>', i); i = e + 1; return n; }
            auto k = element();
            if (!k) return nullptr;
            n->kids.push_back(std::move(k));
        }
    }
};

// ---- Mitsuba import -----------------------------------------------------------------------------------------
static std::vector<float> parse_floats(const std::string &v) {
    std::vector<float> out; std::string t = v;
    for (auto &c : t) if (c == ',') c = ' ';
    std::istringstream is(t); std::string tok;
    while (is >> tok) out.push_back(std::strtof(tok.c_str(), nullptr)); // Rust str::parse::<f32>: correctly rounded, like strtof
    return out;
}
static void transform_point(const float m[16], const float p[3], float out[3]) { for (int r = 0; r < 3; ++r) out[r] = ((m[4 * r] * p[0] + m[4 * r + 1] * p[1]) + m[4 * r + 2] * p[2]) + m[4 * r + 3]; }
static void transform_vector(const float m[16], const float v[3], float out[3]) { for (int r = 0; r < 3; ++r) out[r] = (m[4 * r] * v[0] + m[4 * r + 1] * v[1]) + m[4 * r + 2] * v[2]; }

static int32_t add_const_tex(RenderScene &s, int channels, float a, float b, float c) {
    PtrsTexture t{}; t.kind = PTRS_TEX_CONSTANT; t.channels = channels; t.value[0] = a; t.value[1] = b; t.value[2] = c; t.su = t.sv = 1.0f;
    s.textures.push_back(t);
    return (int32_t)s.textures.size() - 1;
}
static int32_t add_material(RenderScene &s, int kind, std::initializer_list<int32_t> tex, int flags = 0) {
    PtrsMaterial m{}; m.kind = kind; m.flags = flags; m.inner = -1;
    for (int k = 0; k < 6; ++k) m.tex[k] = -1;
    int k = 0; for (int32_t t : tex) m.tex[k++] = t;
    s.materials.push_back(m);
    return (int32_t)s.materials.size() - 1;
}
static std::string snake(const std::string &s) { std::string o; for (char c : s) { if (std::isupper((unsigned char)c)) { o += '_'; o += (char)std::tolower(c); } else o += c; } return o; }

// material_from_bsdf, pathtracer/importer/mitsuba.rs:84-181
static int32_t material_from_bsdf(RenderScene &s, const Node &el, std::string &err) {
    const std::string kind = el.get("type");
    std::map<std::string, std::vector<float>> rgbs; std::map<std::string, float> floats;
    for (auto &k : el.kids) {
        if (k->name == "rgb") rgbs[k->get("name")] = parse_floats(k->get("value"));
        if (k->name == "float") floats[snake(k->get("name"))] = std::strtof(k->get("value").c_str(), nullptr);
    }
    auto rgb = [&](const char *a, const char *b) -> std::vector<float> { if (rgbs.count(a)) return rgbs[a]; if (b && rgbs.count(b)) return rgbs[b]; return {1.0f, 1.0f, 1.0f}; };
    if (kind == "twosided") { const Node *in = el.child("bsdf"); if (!in) { err = "twosided without bsdf"; return -1; } return material_from_bsdf(s, *in, err); }
    if (kind == "diffuse") { auto c = rgb("reflectance", nullptr); return add_material(s, PTRS_MAT_MATTE, {add_const_tex(s, 3, c[0], c[1], c[2])}); }
    if (kind == "conductor" || kind == "roughconductor") {
        const Node *mat = el.child_named("string", "material");
        if (kind == "conductor" && mat) { if (mat->get("value") == "none") return add_material(s, PTRS_MAT_MIRROR, {}); err = "other material values not supported yet!"; return -1; }
        if (!rgbs.count("eta") || !rgbs.count("k")) { err = "conductor without eta/k"; return -1; }
        auto e = rgbs["eta"], k = rgbs["k"], r = rgb("specularReflectance", "specular_reflectance");
        float alpha = kind == "conductor" ? 0.001f : floats["alpha"];
        int32_t t0 = add_const_tex(s, 3, e[0], e[1], e[2]), t1 = add_const_tex(s, 3, k[0], k[1], k[2]), t2 = add_const_tex(s, 3, r[0], r[1], r[2]), t3 = add_const_tex(s, 1, alpha, 0, 0);
        return add_material(s, PTRS_MAT_METAL, {t0, t1, t2, t3, -1, -1}, 0);
    }
    if (kind == "dielectric") {
        int32_t t0 = add_const_tex(s, 3, 1, 1, 1), t1 = add_const_tex(s, 3, 1, 1, 1), t2 = add_const_tex(s, 1, floats["int_ior"], 0, 0);
        return add_material(s, PTRS_MAT_GLASS, {t0, t1, t2});
    }
    if (kind == "plastic" || kind == "roughplastic") {
        float e = floats["int_ior"];
        float r0 = ((e - 1.0f) * (e - 1.0f)) / ((e + 1.0f) * (e + 1.0f));
        float a = kind == "plastic" ? 0.001f : floats["alpha"];
        auto kd = rgb("diffuseReflectance", "diffuse_reflectance");
        int32_t t0 = add_const_tex(s, 3, kd[0], kd[1], kd[2]), t1 = add_const_tex(s, 3, r0, r0, r0), t2 = add_const_tex(s, 1, a, 0, 0), t3 = add_const_tex(s, 1, a, 0, 0);
        return add_material(s, PTRS_MAT_SUBSTRATE, {t0, t1, t2, t3}, 0);
    }
    err = "unsupported bsdf type " + kind;
    return -1;
}

// genmesh 0.6.2 Plane::new / Cube::new (restated from memory)
static void gen_rectangle(std::vector<float> &pos, std::vector<float> &nrm, std::vector<uint32_t> &idx) {
    pos = {-1, -1, 0, 1, -1, 0, -1, 1, 0, 1, 1, 0};
    nrm = {0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};
    idx = {0, 1, 3, 0, 3, 2};
}
static void gen_cube(std::vector<float> &pos, std::vector<float> &nrm, std::vector<uint32_t> &idx) {
    static const float N[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    static const int Q[6][4] = {{6, 7, 5, 4}, {0, 1, 3, 2}, {3, 7, 6, 2}, {4, 5, 1, 0}, {5, 7, 3, 1}, {0, 2, 6, 4}};
    pos.clear(); nrm.clear(); idx.clear();
    for (int f = 0; f < 6; ++f) {
        for (int v = 0; v < 4; ++v) {
            int id = Q[f][v];
            pos.push_back(id & 4 ? 1.0f : -1.0f); pos.push_back(id & 2 ? 1.0f : -1.0f); pos.push_back(id & 1 ? 1.0f : -1.0f);
            nrm.push_back(N[f][0]); nrm.push_back(N[f][1]); nrm.push_back(N[f][2]);
        }
        uint32_t b = 4u * f;
        idx.insert(idx.end(), {b, b + 1, b + 2, b, b + 2, b + 3});
    }
}

// get_camera (common/importer/mitsuba.rs:685-710, Q30) + Camera::new (common/mod.rs:33-62)
static void make_camera(const float cam_to_world[16], float fov_deg, int film_w, int film_h, int res_w, int res_h, Camera &cam) {
    const float PI = 3.14159274101257324f;
    const float fov = fov_deg * (float)(3.14159265358979323846 / 180.0); // f32::to_radians
    // Rotation3::new((0,-pi,0)) = from_axis_angle(normalize, |.|)
    float a[3] = {0.0f, -PI, 0.0f};
    float angle = std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
    float ux = a[0] / angle, uy = a[1] / angle, uz = a[2] / angle;
    float sn = (float)std::sin((double)angle), cs = (float)std::cos((double)angle), omc = 1.0f - cs;
    float sqx = ux * ux, sqy = uy * uy, sqz = uz * uz;
    float R[16] = {sqx + (1.0f - sqx) * cs, ux * uy * omc - uz * sn, ux * uz * omc + uy * sn, 0,
                   ux * uy * omc + uz * sn, sqy + (1.0f - sqy) * cs, uy * uz * omc - ux * sn, 0,
                   ux * uz * omc - uy * sn, uy * uz * omc + ux * sn, sqz + (1.0f - sqz) * cs, 0, 0, 0, 0, 1};
    float M[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float acc = 0.0f; for (int k = 0; k < 4; ++k) acc = acc + cam_to_world[4 * i + k] * R[4 * k + j]; M[4 * i + j] = acc; }
    // try_convert::<Projective3, Similarity3>: normalise columns; scale forced to 1
    float r[3][3];
    for (int c = 0; c < 3; ++c) {
        float x = M[c], y = M[4 + c], z = M[8 + c];
        float n = std::sqrt((x * x + y * y) + z * z);
        r[0][c] = x / n; r[1][c] = y / n; r[2][c] = z / n;
    }
    // UnitQuaternion::from_rotation_matrix
    float tr = (r[0][0] + r[1][1]) + r[2][2], q = 0.25f, w, i, j, k;
    if (tr > 0.0f) { float d = std::sqrt(tr + 1.0f) * 2.0f; w = q * d; i = (r[2][1] - r[1][2]) / d; j = (r[0][2] - r[2][0]) / d; k = (r[1][0] - r[0][1]) / d; }
    else if (r[0][0] > r[1][1] && r[0][0] > r[2][2]) { float d = std::sqrt(((1.0f + r[0][0]) - r[1][1]) - r[2][2]) * 2.0f; w = (r[2][1] - r[1][2]) / d; i = q * d; j = (r[0][1] + r[1][0]) / d; k = (r[0][2] + r[2][0]) / d; }
    else if (r[1][1] > r[2][2]) { float d = std::sqrt(((1.0f + r[1][1]) - r[0][0]) - r[2][2]) * 2.0f; w = (r[0][2] - r[2][0]) / d; i = (r[0][1] + r[1][0]) / d; j = q * d; k = (r[1][2] + r[2][1]) / d; }
    else { float d = std::sqrt(((1.0f + r[2][2]) - r[0][0]) - r[1][1]) * 2.0f; w = (r[1][0] - r[0][1]) / d; i = (r[0][2] + r[2][0]) / d; j = (r[1][2] + r[2][1]) / d; k = q * d; }
    const float rot[4] = {i, j, k, w}, trans[3] = {M[3], M[7], M[11]};
    make_camera_perspective(rot, trans, (float)res_w / (float)res_h, fov * ((float)film_h / (float)film_w), 0.01f, 10000.0f, res_w, res_h, cam);
}

void make_camera_perspective(const float rot[4], const float trans[3], float aspect, float fovy, float zn, float zf, int res_w, int res_h, Camera &cam) {
    PtrsCamera &c = cam.abi;
    std::memcpy(c.rot, rot, 16); std::memcpy(c.trans, trans, 12);
    const float W = (float)res_w, H = (float)res_h;
    c.m11 = 1.0f / (float)std::tan((double)(fovy / 2.0f));
    c.m00 = c.m11 / aspect;
    c.m22 = (zf + zn) / (zn - zf);
    c.m23 = zf * zn * 2.0f / (zn - zf);
    const float sx = W * 0.5f, sy = H * -0.5f, ax = 1.0f / sx, by = 1.0f / sy;
    const float r2s[16] = {ax, 0, 0, -1.0f, 0, by, 0, 1.0f, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(c.raster_to_screen, r2s, 64);
    const float n = c.m22 / c.m23, inv00 = 1.0f / c.m00, inv11 = 1.0f / c.m11;
    const float p0[3] = {(inv00 * -1.0f) / n, (inv11 * 1.0f) / n, -1.0f / n};
    const float px[3] = {(inv00 * ax + inv00 * -1.0f) / n, (inv11 * 1.0f) / n, -1.0f / n};
    const float py[3] = {(inv00 * -1.0f) / n, (inv11 * by + inv11 * 1.0f) / n, -1.0f / n};
    for (int t = 0; t < 3; ++t) { c.dx_camera[t] = px[t] - p0[t]; c.dy_camera[t] = py[t] - p0[t]; }
    cam.film = Film(res_w, res_h);
}

bool import_scene(const std::string &path, int res_w, int res_h, Camera &camera, RenderScene &scene, std::string &err, bool default_lights, const std::string &env_map_path) {
    const size_t dot = path.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".gltf" || ext == ".glb") return import_gltf(path, res_w, res_h, default_lights, env_map_path, camera, scene, err); // importer/mod.rs:17-18
    if (ext != ".xml") { err = "unsupported format!"; return false; } // importer/mod.rs:15-23
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    std::string src((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    Parser P(src);
    auto root = P.element();
    if (!root || root->name != "scene") { err = "not a mitsuba scene: " + P.err; return false; }
    const Node *sensor = root->child("sensor");
    if (!sensor) { err = "scene without sensor"; return false; }
    const Node *fovn = sensor->child_named("float", "fov"), *film = sensor->child("film"), *tf = sensor->child("transform");
    if (!fovn || !film || !tf || !tf->child("matrix")) { err = "sensor needs fov, film and toWorld matrix"; return false; }
    const Node *fw = film->child_named("integer", "width"), *fh = film->child_named("integer", "height");
    if (!fw || !fh) { err = "film needs width and height"; return false; }
    auto cm = parse_floats(tf->child("matrix")->get("value"));
    if (cm.size() != 16) { err = "camera matrix needs 16 values"; return false; }
    make_camera(cm.data(), std::strtof(fovn->get("value").c_str(), nullptr), std::atoi(fw->get("value").c_str()), std::atoi(fh->get("value").c_str()), res_w, res_h, camera);
    scene = RenderScene();
    std::map<std::string, int32_t> named;
    for (auto &k : root->kids) if (k->name == "bsdf") { int32_t m = material_from_bsdf(scene, *k, err); if (m < 0) return false; named[k->get("id")] = m; }
    for (auto &k : root->kids) {
        if (k->name != "shape") continue;
        std::vector<float> pos, nrm; std::vector<uint32_t> idx;
        const std::string kind = k->get("type");
        if (kind == "rectangle") gen_rectangle(pos, nrm, idx);
        else if (kind == "cube") gen_cube(pos, nrm, idx);
        else { err = "unsupported shape type " + kind; return false; }
        float M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        if (const Node *t = k->child("transform")) if (const Node *mn = t->child("matrix")) { auto v = parse_floats(mn->get("value")); if (v.size() != 16) { err = "shape matrix needs 16 values"; return false; } std::memcpy(M, v.data(), 64); }
        Mesh mesh; mesh.pos.resize(pos.size()); mesh.normal.resize(nrm.size()); mesh.indices = idx;
        for (size_t v = 0; v < pos.size() / 3; ++v) { transform_point(M, &pos[3 * v], &mesh.pos[3 * v]); transform_vector(M, &nrm[3 * v], &mesh.normal[3 * v]); } // Q15
        if (const Node *ref = k->child("ref")) { auto it = named.find(ref->get("id")); if (it == named.end()) { err = "unknown bsdf id " + ref->get("id"); return false; } mesh.material = it->second; }
        else if (const Node *b = k->child("bsdf")) { mesh.material = material_from_bsdf(scene, *b, err); if (mesh.material < 0) return false; }
        else { err = "either ref exists or embedded bsdf exists"; return false; }
        const uint32_t mi = (uint32_t)scene.meshes.size();
        if (const Node *em = k->child("emitter")) if (em->get("type") == "area") { // one DiffuseAreaLight per triangle (mitsuba.rs:306-323)
            const Node *rad = em->child("rgb");
            auto c = parse_floats(rad ? rad->get("value") : "1 1 1");
            int32_t ke = add_const_tex(scene, 3, c[0], c[1], c[2]);
            for (uint32_t t = 0; t < idx.size() / 3; ++t) { PtrsLight L{}; L.kind = PTRS_LIGHT_AREA; L.mesh = mi; L.tri = t; L.ke_tex = ke; L.lmap_tex = -1; scene.lights.push_back(L); }
        }
        scene.meshes.push_back(std::move(mesh));
    }
    return true;
}

// ---- PNG (zlib) ---------------------------------------------------------------------------------------------
bool write_png_rgba8(const std::string &path, int w, int h, const std::vector<uint8_t> &rgba, std::string &err) {
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 4 + 1));
    for (int y = 0; y < h; ++y) { raw[(size_t)y * (w * 4 + 1)] = 0; std::memcpy(&raw[(size_t)y * (w * 4 + 1) + 1], &rgba[(size_t)y * w * 4], (size_t)w * 4); }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "zlib compress failed"; return false; }
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) { err = "cannot write " + path; return false; }
    auto be32 = [](uint32_t v, uint8_t *o) { o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; };
    auto chunk = [&](const char *tag, const uint8_t *data, uint32_t n) {
        uint8_t hdr[8]; be32(n, hdr); std::memcpy(hdr + 4, tag, 4);
        std::fwrite(hdr, 1, 8, f); if (n) std::fwrite(data, 1, n, f);
        uLong c = crc32(0L, (const Bytef *)tag, 4); if (n) c = crc32(c, data, n);
        uint8_t cb[4]; be32((uint32_t)c, cb); std::fwrite(cb, 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13]; be32((uint32_t)w, ihdr); be32((uint32_t)h, ihdr + 4); ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13); chunk("IDAT", comp.data(), (uint32_t)clen); chunk("IEND", nullptr, 0);
    std::fclose(f);
    return true;
}

// ---- scene dump (for the cross-check against the Python host, tests/test_host_cpp.py) -------------------------
bool dump_scene(const std::string &path, const Camera &cam, const RenderScene &s) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto w32 = [&](uint32_t v) { std::fwrite(&v, 4, 1, f); };
    std::fwrite("PTRSDUMP", 1, 8, f);
    std::fwrite(&cam.abi, sizeof(PtrsCamera), 1, f);
    w32((uint32_t)s.meshes.size());
    for (auto &m : s.meshes) {
        w32((uint32_t)(m.pos.size() / 3)); w32((uint32_t)(m.indices.size() / 3)); w32((uint32_t)m.material); w32(m.normal.empty() ? 0u : 1u);
        std::fwrite(m.pos.data(), 4, m.pos.size(), f); std::fwrite(m.normal.data(), 4, m.normal.size(), f); std::fwrite(m.indices.data(), 4, m.indices.size(), f);
    }
    w32((uint32_t)s.materials.size());
    for (auto &m : s.materials) { w32((uint32_t)m.kind); for (int k = 0; k < 6; ++k) w32((uint32_t)m.tex[k]); w32((uint32_t)m.flags); w32((uint32_t)m.inner); }
    w32((uint32_t)s.textures.size());
    for (auto &t : s.textures) { w32((uint32_t)t.kind); w32((uint32_t)t.channels); std::fwrite(t.value, 4, 3, f); std::fwrite(t.value2, 4, 3, f); }
    w32((uint32_t)s.lights.size());
    for (auto &l : s.lights) { w32((uint32_t)l.kind); w32(l.mesh); w32(l.tri); w32((uint32_t)l.ke_tex); }
    std::fclose(f);
    return true;
}

bool dump_scene_full(const std::string &path, const Camera &cam, const RenderScene &s) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto w32 = [&](uint32_t v) { std::fwrite(&v, 4, 1, f); };
    auto wf = [&](const float *p, size_t n) { if (n) std::fwrite(p, 4, n, f); };
    std::fwrite("PTRSDMP2", 1, 8, f);
    std::fwrite(&cam.abi, sizeof(PtrsCamera), 1, f);
    w32((uint32_t)s.meshes.size());
    for (auto &m : s.meshes) {
        w32((uint32_t)(m.pos.size() / 3)); w32((uint32_t)(m.indices.size() / 3)); w32((uint32_t)m.material); w32((uint32_t)m.alpha_mask_tex);
        w32((m.normal.empty() ? 0u : 1u) | (m.uv.empty() ? 0u : 2u) | (m.tangent.empty() ? 0u : 4u));
        wf(m.pos.data(), m.pos.size()); wf(m.normal.data(), m.normal.size()); wf(m.uv.data(), m.uv.size()); wf(m.tangent.data(), m.tangent.size());
        std::fwrite(m.indices.data(), 4, m.indices.size(), f);
    }
    w32((uint32_t)s.materials.size());
    for (auto &m : s.materials) { w32((uint32_t)m.kind); for (int k = 0; k < 6; ++k) w32((uint32_t)m.tex[k]); w32((uint32_t)m.flags); w32((uint32_t)m.inner); }
    w32((uint32_t)s.textures.size());
    for (auto &t : s.textures) {
        w32((uint32_t)t.kind); w32((uint32_t)t.channels); wf(t.value, 3); wf(t.value2, 3); wf(&t.su, 1); wf(&t.sv, 1); wf(&t.du, 1); wf(&t.dv, 1);
        w32((uint32_t)t.wrap); w32((uint32_t)t.n_levels);
        for (int l = 0; l < t.n_levels; ++l) { w32((uint32_t)t.level_cols[l]); w32((uint32_t)t.level_rows[l]); wf(t.level_data[l], (size_t)t.level_cols[l] * t.level_rows[l] * t.channels); }
    }
    w32((uint32_t)s.lights.size());
    for (auto &l : s.lights) {
        w32((uint32_t)l.kind); wf(l.v, 3); wf(l.c, 3); w32(l.mesh); w32(l.tri); w32((uint32_t)l.ke_tex); w32((uint32_t)l.lmap_tex);
        wf(l.light_to_world, 16); wf(l.world_to_light, 16); w32((uint32_t)l.dist_nu); w32((uint32_t)l.dist_nv);
        if (l.kind == PTRS_LIGHT_INFINITE) {
            const size_t nu = (size_t)l.dist_nu, nv = (size_t)l.dist_nv;
            wf(l.dist_func, nu * nv); wf(l.dist_cdf, (nu + 1) * nv); wf(l.dist_func_int, nv); wf(l.marg_cdf, nv + 1); wf(&l.marg_func_int, 1);
        }
    }
    std::fclose(f);
    return true;
}

} // namespace ptrs_host
