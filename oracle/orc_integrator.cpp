// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_math.h header).  Parity unpinned.
// Restates src/pathtracer/integrator.rs (estimate_direct 23-139, uniform_sample_one_light
// 192-217, li 392-503, render 536-642, render_single_pixel 505-534), pathtracer/mod.rs:43-81
// (camera rays), common/film.rs + filter.rs (film, Gaussian filter table), and exports a C API
// (oracle.h) used by tests/ through ctypes and by bench.py's cpu_baseline leg.
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include "oracle.h"
#include "orc_bsdf.h"
#include "orc_sampler.h"
#include "orc_stratified.h"

namespace orc {

static thread_local std::string g_err;
static SobolTables g_tables;
static bool g_tables_ok = false;

// ---- camera: pathtracer/mod.rs:59-81 ------------------------------------------------------------
static RayDifferential generate_ray_differential(const PtrsCamera &c, Vec2 p_film) {
    // raster_to_screen * Point3(px, py, 0)  (Affine3 * Point3)
    Vec3 s = affine_point(c.raster_to_screen, Vec3(p_film.x, p_film.y, 0.0f));
    // Perspective3::unproject_point (nalgebra 0.32.2)
    float inverse_denom = c.m23 / (s.z + c.m22);
    Vec3 p_camera(s.x * inverse_denom / c.m00, s.y * inverse_denom / c.m11, -inverse_denom);
    Vec3 trans(c.trans[0], c.trans[1], c.trans[2]);
    Vec3 world_orig = quat_rotate(c.rot, Vec3(0, 0, 0)) + trans; // Isometry3 * Point3
    Vec3 world_dir = quat_rotate(c.rot, p_camera);               // Isometry3 * Vector3
    Vec3 dxc(c.dx_camera[0], c.dx_camera[1], c.dx_camera[2]), dyc(c.dy_camera[0], c.dy_camera[1], c.dy_camera[2]);
    Vec3 rx_world_dir = quat_rotate(c.rot, p_camera + dxc);
    Vec3 ry_world_dir = quat_rotate(c.rot, p_camera + dyc);
    RayDifferential rd;
    rd.ray.o = world_orig; rd.ray.d = normalize(world_dir); rd.ray.t_max = kInf;
    rd.has_differentials = true;
    rd.rx_origin = world_orig; rd.ry_origin = world_orig;
    rd.rx_direction = normalize(rx_world_dir); rd.ry_direction = normalize(ry_world_dir);
    return rd;
}

// ---- film: common/film.rs, filter.rs --------------------------------------------------------------
static constexpr int FILTER_TABLE_WIDTH = 16;
struct Film {
    int W = 0, H = 0;
    float radius = 2.0f;
    float table[FILTER_TABLE_WIDTH * FILTER_TABLE_WIDTH];
    Film(int w, int h) : W(w), H(h) {
        // GuassianFilter::new(2.) filter.rs:67-80 ; Film::new film.rs:133-144
        const float alpha = 2.0f;
        const float expv = pt_expf(-alpha * radius * radius);
        auto gaussian = [&](float d) { return fmax_rs(0.0f, pt_expf(-alpha * d * d) - expv); };
        int off = 0;
        for (int y = 0; y < FILTER_TABLE_WIDTH; y++)
            for (int x = 0; x < FILTER_TABLE_WIDTH; x++) {
                float px = ((float)x + 0.5f) * radius / (float)FILTER_TABLE_WIDTH;
                float py = ((float)y + 0.5f) * radius / (float)FILTER_TABLE_WIDTH;
                table[off++] = gaussian(px) * gaussian(py);
            }
    }
    Bounds2i sample_bounds() const { // film.rs:174-185
        Bounds2i b;
        b.min_x = (int32_t)std::floor(0.5f - radius); b.min_y = (int32_t)std::floor(0.5f - radius);
        b.max_x = (int32_t)std::ceil((float)W - 0.5f + radius); b.max_y = (int32_t)std::ceil((float)H - 0.5f + radius);
        return b;
    }
};

struct FilmTile { // film.rs:24-111
    Bounds2i pb; std::vector<float> px; // rgb + weight per pixel
    const Film *film;
    FilmTile(const Film &f, Bounds2i sb) : film(&f) { // Film::get_film_tile 193-211
        pb.min_x = std::max((int32_t)std::ceil((float)sb.min_x - 0.5f - f.radius), 0);
        pb.min_y = std::max((int32_t)std::ceil((float)sb.min_y - 0.5f - f.radius), 0);
        pb.max_x = std::min((int32_t)std::floor((float)sb.max_x - 0.5f + f.radius) + 1, f.W);
        pb.max_y = std::min((int32_t)std::floor((float)sb.max_y - 0.5f + f.radius) + 1, f.H);
        int w = std::max(pb.max_x - pb.min_x, 0), h = std::max(pb.max_y - pb.min_y, 0);
        px.assign((size_t)w * h * 4, 0.0f);
    }
    void add_sample(Vec2 p_film, Spectrum l) { // 60-106
        float r = film->radius, inv_r = 1.0f / r;
        Vec2 pd{p_film.x - 0.5f, p_film.y - 0.5f};
        int32_t p0x = (int32_t)std::ceil(pd.x - r), p0y = (int32_t)std::ceil(pd.y - r);
        int32_t p1x = (int32_t)(std::floor(pd.x + r) + 1.0f), p1y = (int32_t)(std::floor(pd.y + r) + 1.0f);
        p0x = std::max(p0x, pb.min_x); p0y = std::max(p0y, pb.min_y);
        p1x = std::min(p1x, pb.max_x); p1y = std::min(p1y, pb.max_y);
        int width = pb.max_x - pb.min_x;
        for (int32_t y = p0y; y < p1y; y++) {
            float fy = std::fabs(((float)y - pd.y) * inv_r * (float)FILTER_TABLE_WIDTH);
            int iy = std::min((int)std::floor(fy), FILTER_TABLE_WIDTH - 1);
            for (int32_t x = p0x; x < p1x; x++) {
                float fx = std::fabs(((float)x - pd.x) * inv_r * (float)FILTER_TABLE_WIDTH);
                int ix = std::min((int)std::floor(fx), FILTER_TABLE_WIDTH - 1);
                float w = film->table[iy * FILTER_TABLE_WIDTH + ix];
                float *p = &px[((size_t)(y - pb.min_y) * width + (x - pb.min_x)) * 4];
                p[0] += l.r * w; p[1] += l.g * w; p[2] += l.b * w; p[3] += w;
            }
        }
    }
    void merge_into(PtrsFilmPixel *film_px, int W) const { // Film::merge_film_tile 213-228
        int width = pb.max_x - pb.min_x;
        for (int32_t x = pb.min_x; x < pb.max_x; x++)
            for (int32_t y = pb.min_y; y < pb.max_y; y++) {
                const float *p = &px[((size_t)(y - pb.min_y) * width + (x - pb.min_x)) * 4];
                PtrsFilmPixel &m = film_px[(size_t)y * W + x];
                m.rgb[0] += p[0]; m.rgb[1] += p[1]; m.rgb[2] += p[2]; m.weight += p[3];
            }
    }
};

// ---- integrator -------------------------------------------------------------------------------------
struct Integrator {
    const Scene *scene; int max_depth; float rr_threshold; int rr_start_depth; bool rr_enable;

    // estimate_direct, integrator.rs:23-139 (handle_media = false, specular = false)
    Spectrum estimate_direct(const SurfaceInteraction &it, const BSDF &bsdf, Vec2 u_scattering, uint32_t light_idx, Vec2 u_light, Counters &cnt) const {
        const Light &light = scene->lights[light_idx];
        const uint32_t bsdf_flags = BSDF_ALL & ~BSDF_SPECULAR;
        Spectrum ld(0.0f);
        Vec3 wi; float light_pdf = 0.0f, scattering_pdf = 0.0f;
        Scene::VisibilityTester vis;
        Spectrum li = scene->light_sample_li(light, it.general, u_light, wi, light_pdf, vis);
        if (light_pdf > 0.0f && !li.is_black()) {
            Spectrum f = bsdf.f(it.general.wo, wi, bsdf_flags) * std::fabs(dot(wi, it.shading.n));
            scattering_pdf = bsdf.pdf(it.general.wo, wi, bsdf_flags);
            if (!f.is_black()) {
                if (!scene->unoccluded(vis, &cnt)) li = Spectrum(0.0f);
                if (!li.is_black()) {
                    if (light.is_delta()) ld += f * li / light_pdf;
                    else { float weight = power_heuristic(1, light_pdf, 1, scattering_pdf); ld += f * li * weight / light_pdf; }
                }
            }
        }
        if (!light.is_delta()) {
            uint32_t sampled_type = BSDF_ALL;
            Spectrum f = bsdf.sample_f(it.general.wo, wi, u_scattering, scattering_pdf, bsdf_flags, &sampled_type);
            f *= std::fabs(dot(wi, it.shading.n));
            bool sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
            if (!f.is_black() && scattering_pdf > 0.0f) {
                float weight = 1.0f;
                if (!sampled_specular) {
                    light_pdf = scene->light_pdf_li(light, it.general, wi);
                    if (light_pdf == 0.0f) return ld; // Q11
                    weight = power_heuristic(1, scattering_pdf, 1, light_pdf);
                }
                SurfaceInteraction light_isect;
                Ray ray = it.general.spawn_ray(wi);
                Spectrum tr(1.0f);
                cnt.rays_mis++;
                bool found = scene->intersect(ray, light_isect, &cnt);
                Spectrum li2(0.0f);
                if (found) {
                    // ptr::eq(light, isect_light): the hit primitive's area light is this light (Q11)
                    if (scene->tris[light_isect.prim].area_light == (int32_t)light_idx) li2 = scene->le(light_isect, -wi);
                } else li2 = scene->light_le(light, ray);
                if (!li2.is_black()) ld += f * li2 * tr * weight / scattering_pdf;
            }
        }
        return ld;
    }

    // uniform_sample_one_light, integrator.rs:192-217
    template <class SAMPLER>
    Spectrum uniform_sample_one_light(const SurfaceInteraction &it, const BSDF &bsdf, SAMPLER &sampler, Counters &cnt) const {
        size_t num_lights = scene->lights.size();
        if (num_lights == 0) return Spectrum(0.0f);
        Vec2 u_light = sampler.get_2d();
        Vec2 u_scattering = sampler.get_2d();
        float fl = std::floor(sampler.get_1d() * (float)num_lights);
        size_t light_idx = (size_t)fl; if (light_idx > num_lights - 1) light_idx = num_lights - 1;
        return (float)num_lights * estimate_direct(it, bsdf, u_scattering, (uint32_t)light_idx, u_light, cnt);
    }

    // li, integrator.rs:392-503
    template <class SAMPLER>
    Spectrum li(const RayDifferential &ray_in, SAMPLER &sampler, Counters &cnt) const {
        Spectrum l(0.0f), beta(1.0f);
        RayDifferential ray = ray_in;
        bool specular_bounce = false;
        int32_t bounces = 0;
        float eta_scale = 1.0f;
        while (true) {
            SurfaceInteraction isect;
            cnt.rays_extension++;
            bool found = scene->intersect(ray.ray, isect, &cnt);
            if (bounces == 0 || specular_bounce) {
                if (found) l += beta * scene->le(isect, -ray.ray.d);
                else for (uint32_t li_ : scene->infinite_lights) l += beta * scene->light_le(scene->lights[li_], ray.ray);
            }
            if (!found || bounces >= max_depth) break;
            // SurfaceMediumInteraction::compute_scattering_functions, interaction.rs:283-295
            if (!isect.compute_differentials(ray)) { isect.dudx = isect.dvdx = isect.dudy = isect.dvdy = 0.0f; isect.dpdx = Vec3(); isect.dpdy = Vec3(); }
            BSDF bsdf;
            bool has_bsdf = compute_scattering_functions(*scene, scene->meshes[scene->tris[isect.prim].mesh].material, isect, bsdf);
            if (!has_bsdf) { // Q7
                ray = RayDifferential(isect.general.spawn_ray(ray.ray.d));
                bounces -= 1;
                continue;
            }
            if (bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
                Spectrum ld = beta * uniform_sample_one_light(isect, bsdf, sampler, cnt);
                l += ld;
            }
            Vec3 wo = -ray.ray.d, wi;
            float pdf = 0.0f; uint32_t flags = 0;
            Spectrum f = bsdf.sample_f(wo, wi, sampler.get_2d(), pdf, BSDF_ALL, &flags);
            if (f.is_black() || pdf == 0.0f) break;
            beta *= f * std::fabs(dot(wi, isect.shading.n)) / pdf;
            specular_bounce = (flags & BSDF_SPECULAR) != 0;
            if ((flags & BSDF_SPECULAR) && (flags & BSDF_TRANSMISSION)) {
                float eta = bsdf.eta;
                eta_scale *= dot(wo, isect.general.n) > 0.0f ? eta * eta : 1.0f / (eta * eta);
            }
            ray = RayDifferential(isect.general.spawn_ray(wi));
            if (rr_enable) {
                Spectrum rr_beta = beta * eta_scale;
                if (rr_beta.max_component_value() < rr_threshold && bounces > rr_start_depth) {
                    float q = fmax_rs(0.05f, 1.0f - rr_beta.max_component_value());
                    if (sampler.get_1d() < q) break;
                    beta /= 1.0f - q;
                }
            }
            bounces += 1;
        }
        return l;
    }
};

struct SceneHandle { Scene scene; };

static bool ensure_tables() {
    if (g_tables_ok) return true;
    g_err = "sobol tables not loaded: call orc_load_tables first";
    return false;
}

static Integrator make_integrator(const Scene &s, const PtrsRenderParams &p) {
    Integrator I; I.scene = &s; I.max_depth = p.max_depth; I.rr_threshold = p.rr_threshold; I.rr_start_depth = p.rr_start_depth; I.rr_enable = p.rr_enable != 0;
    return I;
}

} // namespace orc

using namespace orc;

extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }

int orc_load_tables(const char *path) {
    g_tables_ok = g_tables.load(path);
    if (!g_tables_ok) { g_err = std::string("cannot read sobol tables: ") + path; return PTRS_ERR_IO; }
    return PTRS_OK;
}

int orc_scene_create(const PtrsSceneDesc *desc, OrcScene **out) {
    auto *h = new SceneHandle();
    std::string err;
    int rc = h->scene.from_desc(*desc, err);
    if (rc != PTRS_OK) { g_err = err; delete h; return rc; }
    *out = reinterpret_cast<OrcScene *>(h);
    return PTRS_OK;
}
void orc_scene_destroy(OrcScene *s) { delete reinterpret_cast<SceneHandle *>(s); }

int orc_scene_info(OrcScene *s, uint64_t *n_nodes, uint64_t *max_depth, uint64_t *n_tris) {
    Scene &sc = reinterpret_cast<SceneHandle *>(s)->scene;
    if (n_nodes) *n_nodes = sc.nodes.size();
    if (max_depth) *max_depth = sc.bvh_max_depth;
    if (n_tris) *n_tris = sc.tris.size();
    return PTRS_OK;
}

int orc_scene_get_bvh(OrcScene *s, PtrsBvhNode *nodes_out, uint32_t *prims_out) {
    Scene &sc = reinterpret_cast<SceneHandle *>(s)->scene;
    for (size_t i = 0; i < sc.nodes.size(); i++) {
        PtrsBvhNode &n = nodes_out[i];
        for (int k = 0; k < 3; k++) { n.p_min[k] = sc.nodes[i].bounds.p_min[k]; n.p_max[k] = sc.nodes[i].bounds.p_max[k]; }
        n.offset = sc.nodes[i].offset; n.num_prims = sc.nodes[i].num_prims; n.axis = sc.nodes[i].axis; n.pad = 0;
    }
    for (size_t i = 0; i < sc.ordered_prims.size(); i++) prims_out[i] = sc.ordered_prims[i];
    return PTRS_OK;
}

// render: integrator.rs:536-642.  n_threads <= 1: tiles in the serial order of
// cartesian_product(0..nx, 0..ny) (feature disable_rayon); otherwise a dynamic tile queue (rayon).
// sample_rgb (optional): radiance of every sample, layout as ptrs_render_samples.
int orc_render(OrcScene *s, const PtrsCamera *cam, const PtrsRenderParams *p, PtrsFilmPixel *film_px, float *sample_rgb, int n_threads, PtrsStats *stats) {
    if (!ensure_tables()) return PTRS_ERR_INVALID;
    Scene &sc = reinterpret_cast<SceneHandle *>(s)->scene;
    auto t0 = std::chrono::steady_clock::now();
    Film film(p->width, p->height);
    Bounds2i sb = film.sample_bounds();
    const int TILE = 16;
    int ext_x = sb.max_x - sb.min_x, ext_y = sb.max_y - sb.min_y;
    int ntx = (ext_x + TILE - 1) / TILE, nty = (ext_y + TILE - 1) / TILE;
    SobolSampler proto; proto.configure(&g_tables, (size_t)p->spp, sb);
    const bool strat = p->sampler == PTRS_SAMPLER_STRATIFIED;
    size_t strat_dim = 1;
    if (strat) { // StratifiedSamplerBuilder::new(log, dim_pixel_samples, n_sampled_dimensions): spp = dim^2
        while ((strat_dim + 1) * (strat_dim + 1) <= (size_t)std::max(p->spp, 1)) ++strat_dim;
        if (strat_dim * strat_dim != (size_t)p->spp || p->n_sampled_dimensions <= 0) { g_err = "stratified sampler: spp must be a square, n_sampled_dimensions > 0"; return PTRS_ERR_INVALID; }
    }
    const size_t spp = strat ? strat_dim * strat_dim : proto.samples_per_pixel;
    Integrator I = make_integrator(sc, *p);
    // band restriction (multi-GPU rehearsal): only sample rows that can touch output rows [row_begin,row_end)
    int row_b = p->row_begin, row_e = p->row_end;
    if (row_e <= row_b) { row_b = 0; row_e = p->height; }
    std::mutex merge_mu; Counters total; uint64_t n_samples = 0;
    std::atomic<int> next{0};
    std::atomic<bool> strat_overrun{false};
    auto run_tile = [&](auto &sampler, int tile_id, Counters &cnt, uint64_t &ns) {
        int tx = tile_id / nty, ty = tile_id % nty; // (x, y) with x outer
        Bounds2i tb; tb.min_x = sb.min_x + tx * TILE; tb.max_x = std::min(tb.min_x + TILE, sb.max_x);
        tb.min_y = sb.min_y + ty * TILE; tb.max_y = std::min(tb.min_y + TILE, sb.max_y);
        FilmTile tile(film, tb);
        for (int x = tb.min_x; x < tb.max_x; x++)
            for (int y = tb.min_y; y < tb.max_y; y++) {
                sampler.start_pixel(x, y); // (also for pixels outside the band: the stratified sampler's generator runs through the whole tile)
                if (y < row_b - 2 || y >= row_e + 2) continue; // sample rows outside the band's halo
                do {
                    Vec2 p_film = sampler.get_camera_sample(x, y);
                    RayDifferential ray = generate_ray_differential(*cam, p_film);
                    ray.scale_differentials(1.0f / std::sqrt((float)sampler.samples_per_pixel));
                    Spectrum l = I.li(ray, sampler, cnt);
                    ns++;
                    if (sample_rgb) {
                        size_t sidx = ((size_t)(y - sb.min_y) * ext_x + (x - sb.min_x)) * spp + sampler.current_pixel_sample_index;
                        sample_rgb[sidx * 3] = l.r; sample_rgb[sidx * 3 + 1] = l.g; sample_rgb[sidx * 3 + 2] = l.b;
                    }
                    tile.add_sample(p_film, l); // Q24: bad values are still accumulated
                } while (sampler.start_next_sample());
            }
        std::lock_guard<std::mutex> lk(merge_mu);
        // Film::merge_film_tile (film.rs:213-228), restricted to the rows of the band
        int width = tile.pb.max_x - tile.pb.min_x;
        for (int32_t x = tile.pb.min_x; x < tile.pb.max_x; x++)
            for (int32_t y = std::max(tile.pb.min_y, row_b); y < std::min(tile.pb.max_y, row_e); y++) {
                const float *q = &tile.px[((size_t)(y - tile.pb.min_y) * width + (x - tile.pb.min_x)) * 4];
                PtrsFilmPixel &m = film_px[(size_t)y * p->width + x];
                m.rgb[0] += q[0]; m.rgb[1] += q[1]; m.rgb[2] += q[2]; m.weight += q[3];
            }
    };
    auto work = [&](int tile_id, Counters &cnt, uint64_t &ns) {
        if (strat) {
            // integrator.rs:553-554: every tile gets the builder's sampler re-seeded with its index (x-major tile numbering:
            // tile.y * num_tiles.x + tile.x)
            const int tx = tile_id / nty, ty = tile_id % nty;
            StratifiedSampler sampler; sampler.configure(strat_dim, (size_t)p->n_sampled_dimensions, (uint64_t)(ty * ntx + tx));
            run_tile(sampler, tile_id, cnt, ns);
            if (sampler.drew_from_rng) strat_overrun = true;
        } else {
            SobolSampler sampler = proto;
            run_tile(sampler, tile_id, cnt, ns);
        }
    };
    int n_tiles = ntx * nty;
    if (n_threads <= 1) {
        Counters cnt; uint64_t ns = 0;
        for (int t = 0; t < n_tiles; t++) work(t, cnt, ns);
        total.add(cnt); n_samples += ns;
    } else {
        std::vector<std::thread> th;
        std::mutex cm;
        for (int i = 0; i < n_threads; i++)
            th.emplace_back([&]() {
                Counters cnt; uint64_t ns = 0;
                for (;;) { int t = next.fetch_add(1); if (t >= n_tiles) break; work(t, cnt, ns); }
                std::lock_guard<std::mutex> lk(cm); total.add(cnt); n_samples += ns;
            });
        for (auto &t : th) t.join();
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->samples = n_samples; stats->rays_extension = total.rays_extension; stats->rays_shadow = total.rays_shadow; stats->rays_mis = total.rays_mis;
        stats->nodes_visited = total.nodes_visited; stats->tris_tested = total.tris_tested;
        stats->bvh_nodes = sc.nodes.size(); stats->bvh_max_depth = sc.bvh_max_depth;
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (strat_overrun) stats->error_flags |= PTRS_ERRFLAG_SOBOL_DIM; // a path drew past n_sampled_dimensions (straight from the generator)
    }
    return PTRS_OK;
}

// StratifiedSampler's per-pixel tables for one tile (stratified.rs:87-148): the tile's pixels in the reference's order
// (x outer, y inner), for each pixel n_dims x spp 1-D samples followed by n_dims x spp 2-D samples.  Also the raw generator:
// out_u64[n] = the first n outputs of Pcg64Mcg::from_state(state_lo | state_hi << 64) when n_raw > 0.
int orc_stratified_tile(uint64_t seed, int32_t tile_w, int32_t tile_h, int32_t dim_pixel_samples, int32_t n_dims, float *out /* w*h*n_dims*spp*3 */) {
    StratifiedSampler sp; sp.configure((size_t)dim_pixel_samples, (size_t)n_dims, seed);
    const size_t spp = sp.samples_per_pixel;
    size_t o = 0;
    for (int x = 0; x < tile_w; ++x)
        for (int y = 0; y < tile_h; ++y) {
            sp.start_pixel(x, y);
            for (int d = 0; d < n_dims; ++d) for (size_t k = 0; k < spp; ++k) out[o++] = sp.samples_1d[(size_t)d][k];
            for (int d = 0; d < n_dims; ++d) for (size_t k = 0; k < spp; ++k) { out[o++] = sp.samples_2d[(size_t)d][k].x; out[o++] = sp.samples_2d[(size_t)d][k].y; }
        }
    return PTRS_OK;
}
void orc_pcg64mcg(uint64_t state_lo, uint64_t state_hi, uint32_t n, uint64_t *out) {
    Pcg64Mcg r = Pcg64Mcg::from_state((unsigned __int128)state_lo | ((unsigned __int128)state_hi << 64));
    for (uint32_t i = 0; i < n; ++i) out[i] = r.next_u64();
}

// render_single_pixel, integrator.rs:505-534
int orc_render_single_pixel(OrcScene *s, const PtrsCamera *cam, const PtrsRenderParams *p, int32_t px, int32_t py, float *rgb_out) {
    if (!ensure_tables()) return PTRS_ERR_INVALID;
    Scene &sc = reinterpret_cast<SceneHandle *>(s)->scene;
    Film film(p->width, p->height);
    SobolSampler sampler; sampler.configure(&g_tables, (size_t)p->spp, film.sample_bounds());
    Integrator I = make_integrator(sc, *p);
    Counters cnt;
    sampler.start_pixel(px, py);
    do {
        Vec2 p_film = sampler.get_camera_sample(px, py);
        RayDifferential ray = generate_ray_differential(*cam, p_film);
        ray.scale_differentials(1.0f / std::sqrt((float)sampler.samples_per_pixel));
        Spectrum l = I.li(ray, sampler, cnt);
        size_t i = sampler.current_pixel_sample_index;
        rgb_out[3 * i] = l.r; rgb_out[3 * i + 1] = l.g; rgb_out[3 * i + 2] = l.b;
    } while (sampler.start_next_sample());
    return PTRS_OK;
}

int orc_trace_rays(OrcScene *s, uint32_t n, const float *rays, int32_t any_hit, int32_t brute_force, PtrsHit *hits, PtrsStats *stats) {
    Scene &sc = reinterpret_cast<SceneHandle *>(s)->scene;
    Counters cnt;
    for (uint32_t i = 0; i < n; i++) {
        Ray r; r.o = Vec3(rays[7 * i], rays[7 * i + 1], rays[7 * i + 2]); r.d = Vec3(rays[7 * i + 3], rays[7 * i + 4], rays[7 * i + 5]); r.t_max = rays[7 * i + 6];
        PtrsHit &h = hits[i]; h.prim = -1; h.t = r.t_max; h.b0 = h.b1 = h.b2 = 0.0f;
        if (any_hit) { h.prim = sc.intersect_p(r, &cnt) ? 0 : -1; continue; }
        SurfaceInteraction si;
        bool found = brute_force ? sc.intersect_brute(r, si) : sc.intersect(r, si, &cnt);
        if (found) {
            h.prim = si.prim; h.t = r.t_max;
            h.b0 = si.bary[0]; h.b1 = si.bary[1]; h.b2 = si.bary[2];
        }
    }
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->nodes_visited = cnt.nodes_visited; stats->tris_tested = cnt.tris_tested; }
    return PTRS_OK;
}

int orc_sobol_samples(const PtrsRenderParams *p, uint32_t n, const int32_t *px, const int32_t *py, const uint64_t *sample_nums, const uint32_t *dims, float *out, uint64_t *index_out) {
    if (!ensure_tables()) return PTRS_ERR_INVALID;
    Film film(p->width, p->height);
    SobolSampler s; s.configure(&g_tables, (size_t)p->spp, film.sample_bounds());
    for (uint32_t i = 0; i < n; i++) {
        s.start_pixel(px[i], py[i]);
        int64_t idx = s.get_index_for_sample(sample_nums[i]);
        if (index_out) index_out[i] = (uint64_t)idx;
        out[i] = s.sample_dimension(idx, dims[i]);
    }
    return PTRS_OK;
}

int orc_filter_table(float *out256) { Film f(16, 16); std::memcpy(out256, f.table, sizeof(f.table)); return PTRS_OK; }

// known-answer helpers for tests/test_oracle_math.py (reference tests common/math.rs:264-299)
uint32_t orc_log2_int(uint64_t v) { return log2_int(v); }
int orc_solve_2x2(const float *a, const float *b, float *x) { return solve_linear_system_2x2(a, b, x) ? 1 : 0; }
float orc_next_float_up(float v) { return next_float_up(v); }
float orc_next_float_down(float v) { return next_float_down(v); }
float orc_detmath(int fn, float x, float y) {
    switch (fn) {
        case 0: return pt_sinf(x); case 1: return pt_cosf(x); case 2: return pt_logf(x); case 3: return pt_log2f(x);
        case 4: return pt_expf(x); case 5: return pt_powf(x, y); case 6: return pt_atan2f(x, y); case 7: return pt_acosf(x);
        case 9: { float sn, cs; pt_sincosf(x, &sn, &cs); return sn; } case 10: { float sn, cs; pt_sincosf(x, &sn, &cs); return cs; }
        default: return pt_tanf(x);
    }
}

// BxDF grids: evaluate sample_f / f / pdf of a material's BSDF at a synthetic hit with normal +z
// (used to produce and check tests/golden lobe vectors).  mat: one PtrsMaterial with constant
// textures given in tex_values[6][3].
int orc_bsdf_eval(const PtrsMaterial *mat, const float *tex_values, uint32_t n, const float *wo, const float *u, float *out /* n*8: f.rgb, pdf, wi.xyz, flags */) {
    Scene sc;
    sc.textures.resize(6);
    for (int i = 0; i < 6; i++) { sc.textures[i].kind = PTRS_TEX_CONSTANT; for (int c = 0; c < 3; c++) sc.textures[i].value[c] = tex_values[3 * i + c]; }
    Material m; m.kind = mat->kind; m.flags = mat->flags; m.inner = -1;
    for (int i = 0; i < 6; i++) m.tex[i] = mat->tex[i] < 0 ? -1 : i;
    sc.materials.push_back(m);
    for (uint32_t i = 0; i < n; i++) {
        SurfaceInteraction si = SurfaceInteraction::make(Vec3(0, 0, 0), Vec3(), Vec2{0.25f, 0.5f}, Vec3(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]), Vec3(1, 0, 0), Vec3(0, 1, 0), Vec3(), Vec3(), 0);
        BSDF bsdf; float *o = out + 8 * (size_t)i;
        for (int k = 0; k < 8; k++) o[k] = 0.0f;
        if (!compute_scattering_functions(sc, 0, si, bsdf)) { o[7] = -1.0f; continue; }
        Vec3 wi; float pdf = 0.0f; uint32_t flags = 0;
        Spectrum f = bsdf.sample_f(si.general.wo, wi, Vec2{u[2 * i], u[2 * i + 1]}, pdf, BSDF_ALL, &flags);
        o[0] = f.r; o[1] = f.g; o[2] = f.b; o[3] = pdf; o[4] = wi.x; o[5] = wi.y; o[6] = wi.z; o[7] = (float)flags;
    }
    return PTRS_OK;
}

} // extern "C"
