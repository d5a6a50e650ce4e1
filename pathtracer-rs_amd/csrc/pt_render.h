// pt_render.h -- orchestration of PathIntegrator::render (src/pathtracer/integrator.rs:536-642)
// as a wavefront pipeline, independent of where the stages execute.
//
// The reference walks 16x16 tiles with rayon; here the unit of scheduling is a *pass*: a block of
// sample rows x a block of sample indices, whose paths are all in flight at once (sized to use
// HBM generously: ~300 bytes of state per path).  Within a pass the reference's per-path loop
// (integrator.rs:406-500) becomes `max_depth + 1` rounds of
//     extend (trace + bucketing) -> shade[material] -> connect (shadow / MIS queries + resolve)
// and the film is updated once per pass by a deterministic gather.  The sampler ignores the tile
// seed (sobol.rs:75-77), so the decomposition does not change any sample value (Q3).
//
// `BE` is the execution back end: the HIP back end (ptrs_hip.hip) launches gfx950 kernels; the
// test-only host twin (tests/host_twin) runs the same stage functions serially on the CPU.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>

#include "pt_host_scene.h"
#include "pt_items.h"

namespace pt {

struct SobolTablesHost {
    std::vector<uint32_t> matrices;
    std::vector<uint64_t> vdc, vdc_inv;
    uint32_t stride = 52;
    bool ok = false;
};

inline bool load_sobol_tables(const char *path, SobolTablesHost &T) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    char magic[8]; uint32_t hdr[6]; uint32_t lens[52];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTRSSOB1", 8) == 0 && std::fread(hdr, 4, 6, f) == 6 && hdr[0] == 1024 && hdr[1] == 52;
    if (ok) {
        T.stride = hdr[4];
        T.matrices.resize(1024u * 52u); T.vdc.resize((size_t)hdr[2] * T.stride); T.vdc_inv.resize((size_t)hdr[3] * T.stride);
        ok = std::fread(T.matrices.data(), 4, T.matrices.size(), f) == T.matrices.size() && std::fread(lens, 4, 52, f) == 52 &&
             std::fread(T.vdc.data(), 8, T.vdc.size(), f) == T.vdc.size() && std::fread(T.vdc_inv.data(), 8, T.vdc_inv.size(), f) == T.vdc_inv.size();
    }
    std::fclose(f);
    T.ok = ok;
    return ok;
}

// bytetab[dim][k][b] = XOR of SOBOL_MATRICES_32[dim*52 + 8k + j] over the set bits j of b (columns >= 52 do not exist: an
// index never has those bits, 2m + log2(spp) <= 62 is checked by render_impl)
inline void build_sobol_bytetab(const SobolTablesHost &T, std::vector<uint32_t> &out) {
    out.assign((size_t)SOBOL_TAB_DIMS * 8 * 256, 0u);
    for (uint32_t d = 0; d < (uint32_t)SOBOL_TAB_DIMS; ++d)
        for (uint32_t k = 0; k < 8; ++k)
            for (uint32_t b = 0; b < 256; ++b) {
                uint32_t v = 0;
                for (uint32_t j = 0; j < 8; ++j)
                    if ((b >> j) & 1u) { uint32_t col = 8 * k + j; if (col < 52) v ^= T.matrices[(size_t)d * 52 + col]; }
                out[((size_t)d * 8 + k) * 256 + b] = v;
            }
}

inline int32_t round_up_pow2_i32(int32_t v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; }
inline int64_t round_up_pow2_i64(int64_t v) { v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; return v + 1; }

// Film::new filter table with GuassianFilter::new(2.0) (film.rs:133-144, filter.rs:61-90)
inline void gaussian_filter_table(float *t256) {
    const float alpha = 2.0f, radius = 2.0f;
    const float expv = pt_expf(-alpha * radius * radius);
    int off = 0;
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) {
            float px = ((float)x + 0.5f) * radius / 16.0f, py = ((float)y + 0.5f) * radius / 16.0f;
            float gx = max_(0.0f, pt_expf(-alpha * px * px) - expv), gy = max_(0.0f, pt_expf(-alpha * py * py) - expv);
            t256[off++] = gx * gy;
        }
}

struct SampleGrid { int32_t min_x, min_y, NX, NY; uint32_t spp; uint32_t log2_res; int32_t resolution; };

// Film::get_sample_bounds (film.rs:174-185) + SobolSamplerBuilder::new (sobol.rs:35-60)
inline SampleGrid make_sample_grid(int32_t W, int32_t H, int32_t spp_in) {
    SampleGrid g;
    const float r = 2.0f;
    g.min_x = (int32_t)std::floor(0.5f - r); g.min_y = (int32_t)std::floor(0.5f - r);
    int32_t max_x = (int32_t)std::ceil((float)W - 0.5f + r), max_y = (int32_t)std::ceil((float)H - 0.5f + r);
    g.NX = max_x - g.min_x; g.NY = max_y - g.min_y;
    g.spp = (uint32_t)round_up_pow2_i64((int64_t)(spp_in < 1 ? 1 : spp_in));
    g.resolution = round_up_pow2_i32(g.NX > g.NY ? g.NX : g.NY);
    g.log2_res = 31u - (uint32_t)__builtin_clz((uint32_t)g.resolution);
    return g;
}

// FEAT_* bits (pt_texture.h) the shade kernels need for a scene, rounded up to one of the instantiated sets:
// none, image textures, image textures + environment light, everything
inline int scene_features(const HostScene &H) {
    int need = 0;
    if (!H.inf_lights.empty()) need |= FEAT_INFINITE;
    if (H.has_alpha) need |= FEAT_ALPHA;
    for (const DTexture &t : H.texs) if (t.kind == PTRS_TEX_IMAGE) need |= FEAT_IMAGE;
    for (const DMaterial &m : H.mats) if (m.kind == PTRS_MAT_NORMAL) need |= FEAT_NORMAL;
    if (need == 0) return FEAT_SIMPLE;
    if ((need & ~FEAT_IMG) == 0) return FEAT_IMG;
    if ((need & ~FEAT_IMG_ENV) == 0) return FEAT_IMG_ENV;
    return FEAT_FULL;
}
// Feature set the extension / connection kernels need: alpha masks (tested inside the traversal loop: FEAT_FULL), or only textured
// emission / an environment light (read by the epilogue and the MIS resolve behind the loop: FEAT_IMG_ENV, whose traversal loop is the
// lean one), or none of them.  (A scene whose only image textures sit on materials can run the lean traversal kernels.)
inline int scene_trace_features(const HostScene &H) {
    if (H.has_alpha) return FEAT_FULL;
    bool any = !H.inf_lights.empty();
    for (const DLight &L : H.lights) any = any || (L.kind == PTRS_LIGHT_AREA && !L.ke_const);
    return any ? FEAT_IMG_ENV : FEAT_SIMPLE;
}

// Output-row bands for n devices: bounds[0] = 0 <= bounds[1] <= ... <= bounds[n] = height.  Without costs the
// rows are split as evenly as possible (earlier bands take the remainder); with a per-row cost (e.g. rays of a 1-spp
// probe) band k ends at the first row where the running cost reaches k/n of the total -- every band keeps at least one
// row while rows remain.  A band's device also traces the 2-row filter halo on each side (film.rs:60-106), which is
// not part of its cost here.
inline void plan_bands(int32_t height, uint32_t n, const float *row_cost, int32_t *bounds) {
    bounds[0] = 0; bounds[n] = height;
    std::vector<double> pre((size_t)height + 1, 0.0); // pre[y] = cost of rows [0, y)
    if (row_cost) for (int32_t y = 0; y < height; ++y) pre[(size_t)y + 1] = pre[(size_t)y] + (row_cost[y] > 0.0f ? (double)row_cost[y] : 0.0);
    const double total = pre[(size_t)height];
    if (!row_cost || !(total > 0.0)) {
        const int32_t base = height / (int32_t)n, rem = height % (int32_t)n;
        for (uint32_t k = 1; k < n; ++k) bounds[k] = (int32_t)k * base + std::min<int32_t>((int32_t)k, rem);
        return;
    }
    for (uint32_t k = 1; k < n; ++k) {
        const double target = total * (double)k / (double)n;
        int32_t y = (int32_t)(std::upper_bound(pre.begin(), pre.end(), target) - pre.begin()) - 1; // most rows whose cost stays <= target
        // of the two cuts around the target take the closer one
        if (y < height && pre[(size_t)y + 1] - target < target - pre[(size_t)y]) ++y;
        const int32_t lo = std::min<int32_t>(bounds[k - 1] + 1, height);            // at least one row per band while rows remain ...
        const int32_t hi = std::max<int32_t>(lo, height - (int32_t)(n - k));        // ... also for the bands after this one
        bounds[k] = std::min(std::max(y, lo), hi);
    }
}

struct RenderProgress { void (*fn)(void *user, uint32_t done, uint32_t total, int32_t row_begin, int32_t row_end); void *user; };
// ptrs_render_dump_rays: the render stops in its first pass before round `round` and hands out the extension rays of that round
// (round 0: the camera rays; round k: the rays leaving the paths' k-th vertices) -- ray sets of a real frame for the traversal bench.
struct RayDump { uint32_t round, max_rays; float *out /* backend memory, max_rays x 7: o, d, t_max */; uint32_t *n_out /* host */; };

template <class BE>
int render_impl(BE &be, const DScene &sc, const HostScene &sc_host_feat, uint32_t bvh_depth, const PtrsCamera &cam, const PtrsRenderParams &prm,
                v4 *film /* backend memory, W*H */, float *samples_out /* backend memory or null */, PtrsStats *stats, std::string &err,
                const RenderProgress *progress = nullptr /* called after every pass with the rows it touched (back end copies them out first) */,
                const RayDump *dump = nullptr,
                const int32_t *single_pixel = nullptr /* render_single_pixel (integrator.rs:505-534): raster (px, py), any pixel of the sample bounds;
                                                         traces that pixel's spp paths only, no film, samples_out = spp * 3 floats */,
                uint32_t *row_cost = nullptr /* backend memory, NY counters (zeroed by the caller): the BVH queries of every sample row's paths are added; no film */) {
    using clock = std::chrono::steady_clock;
    const bool *kinds_present = sc_host_feat.kinds_present;
    auto t_begin = clock::now();
    if (prm.width <= 0 || prm.height <= 0 || prm.spp <= 0 || prm.max_depth < 0) { err = "bad render parameters"; return PTRS_ERR_INVALID; }
    if (bvh_depth > 64) { err = "BVH deeper than the 64-entry traversal stack (accelerator.rs:370)"; return PTRS_ERR_UNSUPPORTED; }
    SampleGrid g = make_sample_grid(prm.width, prm.height, prm.spp);
    uint32_t strat_dim = 0;
    if (prm.sampler == PTRS_SAMPLER_STRATIFIED) { // StratifiedSamplerBuilder::new(log, dim_pixel_samples, n_sampled_dimensions) (stratified.rs:22-36)
        strat_dim = 1; while ((strat_dim + 1u) * (strat_dim + 1u) <= (uint32_t)prm.spp) ++strat_dim;
        if (strat_dim * strat_dim != (uint32_t)prm.spp) { err = "stratified sampler: spp must be dim_pixel_samples squared"; return PTRS_ERR_INVALID; }
        if (prm.n_sampled_dimensions < 3 * (prm.max_depth + 1) + 1 || prm.n_sampled_dimensions > 63) {
            err = "stratified sampler: n_sampled_dimensions must cover the path (3 * (max_depth + 1) + 1 .. 63): draws past it come from the tile's generator in path order (sampler/mod.rs:137-151), which a wavefront cannot reproduce";
            return PTRS_ERR_UNSUPPORTED;
        }
        if (single_pixel) { err = "render_single_pixel with the stratified sampler is not supported (its tables depend on the tile's earlier pixels)"; return PTRS_ERR_UNSUPPORTED; }
        g.spp = strat_dim * strat_dim;
    } else if (prm.sampler != PTRS_SAMPLER_SOBOL) { err = "unknown sampler"; return PTRS_ERR_INVALID; }
    if (g.log2_res < 1 || g.log2_res > 25 || 2u * g.log2_res + (31u - (uint32_t)__builtin_clz(g.spp)) > 62u) { err = "resolution / spp outside the Sobol index range"; return PTRS_ERR_UNSUPPORTED; }
    int32_t rb = prm.row_begin, re = prm.row_end;
    if (re <= rb) { rb = 0; re = prm.height; }
    if (rb < 0 || re > prm.height) { err = "row band outside the film"; return PTRS_ERR_INVALID; }
    // sample rows (grid coordinates) whose footprint can touch output rows [rb, re)
    int32_t srow0 = std::max(rb, 0), srow1 = std::min(re + 4, g.NY);
    if (single_pixel) {
        const int32_t sx = single_pixel[0] - g.min_x, sy = single_pixel[1] - g.min_y;
        if (sx < 0 || sx >= g.NX || sy < 0 || sy >= g.NY) { err = "pixel outside the sample bounds"; return PTRS_ERR_INVALID; }
        if (!samples_out) { err = "null argument"; return PTRS_ERR_INVALID; }
        srow0 = sy; srow1 = sy + 1;
    }

    DSampler S;
    S.matrices = be.sobol_matrices(); S.bytetab = be.sobol_bytetab(); S.nibtab = be.sobol_nibtab(); S.vdc = be.sobol_vdc(g.log2_res - 1); S.vdc_inv = be.sobol_vdc_inv(g.log2_res - 1);
    S.log2_res = g.log2_res; S.resolution = g.resolution; S.min_x = g.min_x; S.min_y = g.min_y; S.spp = g.spp;
    S.kind = (uint32_t)prm.sampler; S.strat_dims = (uint32_t)prm.n_sampled_dimensions; S.strat1 = nullptr; S.strat2 = nullptr;
    if (strat_dim) { // every tile's generator runs through the whole tile, whichever rows this call renders
        int rc_t = be.strat_tables(g.NX, g.NY, strat_dim, S.strat_dims, &S.strat1, &S.strat2, err);
        if (rc_t != PTRS_OK) return rc_t;
    }
    DCamera C;
    std::memcpy(C.rot, cam.rot, 16); std::memcpy(C.trans, cam.trans, 12);
    C.m00 = cam.m00; C.m11 = cam.m11; C.m22 = cam.m22; C.m23 = cam.m23;
    std::memcpy(C.r2s, cam.raster_to_screen, 64); std::memcpy(C.dxc, cam.dx_camera, 12); std::memcpy(C.dyc, cam.dy_camera, 12);
    DParams R;
    std::memset(&R, 0, sizeof(R));
    R.max_depth = prm.max_depth; R.rr_threshold = prm.rr_threshold; R.rr_start_depth = prm.rr_start_depth; R.rr_enable = prm.rr_enable;
    R.NX = g.NX; R.NY = g.NY; R.W = prm.width; R.H = prm.height;
    R.inv_sqrt_spp = 1.0f / std::sqrt((float)g.spp);
    R.counters_on = (prm.flags & PTRS_FLAG_COUNTERS) ? 1u : 0u;
    R.row_cost = row_cost;

    // ---- pass planning ----------------------------------------------------------------------
    // Passes run on `be.lanes()` independent pipelines (own path state, queues and stream): while one pass is in the thin
    // tail of a kernel or waits for its next launch, the other pass's workgroups fill the machine (two concurrent
    // processes on one MI355X measured +18 % over one).  Film kernels are chained in pass order, so the film is formed
    // by exactly the same additions as with a single pipeline.
    const uint32_t n_lanes = std::max(1u, be.lanes((uint64_t)(srow1 - srow0) * (uint64_t)g.NX * (uint64_t)g.spp, sc_host_feat.kinds_present, single_pixel != nullptr, g.spp)); // (the back end may choose by the size of the job)
    // up to 2^27 paths (~35 GB of path state) per lane: HBM (288 GB) is plentiful, launches and tails are not free.  The back
    // end bounds it by its share of the memory that is free right now, so the library stays embeddable beside other users.
    uint64_t capacity = std::min<uint64_t>(1ull << 27, prm.paths_per_pass ? prm.paths_per_pass : be.auto_capacity(n_lanes, sc_host_feat.kinds_present)); // (2^27: a path slot is 27 bits of a NEE-queue entry, pt_scene.h)
    if (capacity < (uint64_t)g.NX) capacity = (uint64_t)g.NX;
    if (single_pixel) capacity = std::max<uint64_t>(capacity, (uint64_t)g.NX * g.spp); // one pass: the pixel's spp paths (planned below as one row x all samples, launched as spp paths)
    const uint64_t band_rows = (uint64_t)(srow1 - srow0);
    uint64_t rows_per_pass, samples_per_pass;
    if (band_rows * (uint64_t)g.NX <= capacity) {
        rows_per_pass = band_rows;
        // balance the sample chunks: n passes of (almost) equal size instead of full passes plus a small tail
        const uint64_t spp_max = std::max<uint64_t>(1, std::min<uint64_t>(g.spp, capacity / (band_rows * (uint64_t)g.NX)));
        const uint64_t n_chunks = (g.spp + spp_max - 1) / spp_max;
        samples_per_pass = (g.spp + n_chunks - 1) / n_chunks;
    } else {
        samples_per_pass = 1;
        const uint64_t rows_max = std::max<uint64_t>(1, capacity / (uint64_t)g.NX);
        const uint64_t n_chunks = (band_rows + rows_max - 1) / rows_max;
        rows_per_pass = (band_rows + n_chunks - 1) / n_chunks;
    }
    if (n_lanes > 1 && band_rows * (uint64_t)g.NX <= capacity && !prm.paths_per_pass) { // an even number of equal sample chunks
        const uint64_t spp_max = std::max<uint64_t>(1, std::min<uint64_t>(g.spp, capacity / (band_rows * (uint64_t)g.NX)));
        uint64_t n_chunks = (g.spp + spp_max - 1) / spp_max;
        if (g.spp >= n_lanes) n_chunks = ((n_chunks + n_lanes - 1) / n_lanes) * n_lanes;
        samples_per_pass = (g.spp + n_chunks - 1) / n_chunks;
    }
    const uint64_t max_paths = rows_per_pass * (uint64_t)g.NX * samples_per_pass;
    if (max_paths >= 0xffffffffull) { err = "pass too large"; return PTRS_ERR_INVALID; }

    const uint32_t fixed_iters = (uint32_t)prm.max_depth + 1u; // li() runs at most max_depth+1 scene queries (Q7)
    const uint32_t max_iters = fixed_iters + 64u;              // head-room for null-BSDF skips (bounces -= 1)
    const int feat = scene_features(sc_host_feat), feat_trace = scene_trace_features(sc_host_feat);
    int rc = be.begin(sc, S, C, (uint32_t)max_paths, max_iters + 1u, bvh_depth, prm.flags, feat, feat_trace, err);
    if (rc != PTRS_OK) return rc;

    PtrsStats st;
    std::memset(&st, 0, sizeof(st));
    std::vector<uint32_t> counts((size_t)(max_iters + 1u) * Q_STRIDE);
    const uint32_t n_passes_total = (uint32_t)(((band_rows + rows_per_pass - 1) / rows_per_pass) * ((g.spp + samples_per_pass - 1) / samples_per_pass));
    struct Pending { bool active = false, open = false; uint32_t it = 0, n_paths = 0, pass_no = 0; int32_t y0 = 0, y1 = 0; };
    std::vector<Pending> pending(n_lanes);
    bool null_skip_overrun = false;
    auto round = [&](uint32_t i) {
        be.extend(i);                                                     // closest hit + emission / miss / depth cut + material buckets
        be.presample(i);                                                  // (gfx950: the environment light's samples of the round's vertices, ahead of the shade kernels)
        for (int k = 0; k < 6; ++k) if (kinds_present[k]) be.shade(i, k); // one specialised kernel per material bucket
        be.connect(i);                                                    // shadow + MIS queries, resolve into L
    };
    // A path can outlive max_depth + 1 rounds only through null-BSDF skips (`bounces -= 1`, integrator.rs:434-439), which only glass
    // can produce.  Such a pass is enqueued with a few rounds to spare (a round nobody reaches costs its launches: the kernels look at
    // the round's alive flag and leave) and stays OPEN: whether it needs more is asked when its lane is waited for anyway -- before
    // the lane's next pass, or at the end -- not right behind its launch, where the question would stall the host until the pass is
    // through and keep the other lanes empty (round 2: glass scenes ran their lanes one after the other).  The film follows then.
    const uint32_t spare_rounds = kinds_present[PTRS_MAT_GLASS] ? 6u : 0u;
    auto close_pass = [&](uint32_t lane) {
        Pending &pd = pending[lane];
        if (!pd.active || !pd.open) return;
        be.select(lane);
        uint32_t it = pd.it;
        if (spare_rounds) { // (only null-BSDF skips can keep a path alive beyond max_depth + 1 rounds: other scenes are not asked)
            while (it < max_iters && be.read_count(it, Q_EXT) != 0) { round(it); ++it; }
            if (it == max_iters && be.read_count(it, Q_EXT) != 0) null_skip_overrun = true; // paths still alive: never dropped silently
        }
        pd.it = it;
        if (pd.y1 > pd.y0 && film) be.film(film, pd.y0, pd.y1); // ordered after the previous pass's film kernel, whichever lane ran it
        if (samples_out) be.export_samples(samples_out);
        pd.open = false;
    };
    auto finish = [&](uint32_t lane) { // collect the counters of the pass a lane ran last (waits for that lane only)
        Pending &pd = pending[lane];
        if (!pd.active) return;
        close_pass(lane);
        be.select(lane);
        be.read_counts(counts.data(), pd.it + 1u);
        be.learn(counts.data(), std::min(pd.it + 1u, max_iters + 1u), pd.n_paths); // (gfx950: the scene's survival profile, which places the next passes' hand-over to the fused tail)
        for (uint32_t i = 0; i <= pd.it && i < max_iters + 1u; ++i) {
            st.rays_extension += counts[(size_t)i * Q_STRIDE + Q_EXT];
            st.rays_shadow += counts[(size_t)i * Q_STRIDE + Q_SHADOW];
            st.rays_mis += counts[(size_t)i * Q_STRIDE + Q_MIS];
        }
        st.samples += pd.n_paths;
        st.passes += 1;
        // progressive render: the pass's rows are published here, when its lane is waited for anyway (before the lane's next pass, or at
        // the end), not right behind its launch -- the passes on the other lanes keep running meanwhile.  Passes finish in launch order
        // (their film kernels are chained), so the callbacks arrive in pass order.
        if (progress && progress->fn && pd.y1 > pd.y0) { be.publish_rows(film, pd.y0, pd.y1); progress->fn(progress->user, pd.pass_no + 1u, n_passes_total, pd.y0, pd.y1); }
        pd.active = false;
    };
    uint32_t pass_no = 0;
    for (int32_t r0 = srow0; r0 < srow1; r0 += (int32_t)rows_per_pass) {
        const int32_t r1 = std::min<int32_t>(srow1, r0 + (int32_t)rows_per_pass);
        for (uint32_t s0 = 0; s0 < g.spp; s0 += (uint32_t)samples_per_pass, ++pass_no) {
            const uint32_t s1 = (uint32_t)std::min<uint64_t>(g.spp, (uint64_t)s0 + samples_per_pass);
            const uint32_t lane = pass_no % n_lanes;
            // (an open pass is closed -- asked for more rounds, its film kernel enqueued -- when its lane comes round again: passes close in pass
            // order, the order their film kernels are chained in)
            finish(lane);
            be.select(lane);
            R.row0 = r0; R.row1 = r1; R.s0 = s0; R.s1 = s1;
            R.n_paths = (uint32_t)(r1 - r0) * (uint32_t)g.NX * (s1 - s0);
            if (single_pixel) { R.pixel_mode = 1; R.pix_sx = single_pixel[0] - g.min_x; R.pix_sy = single_pixel[1] - g.min_y; R.n_paths = s1 - s0; }
            be.pass_begin(R);
            be.generate();
            uint32_t it = 0;
            if (dump) { // (test / bench hook: nothing is rendered beyond the rounds before the dump)
                for (; it < dump->round && it < max_iters; ++it) round(it);
                be.dump_rays(it, *dump);
                be.end(st);
                if (stats) *stats = st;
                return PTRS_OK;
            }
            // Rounds as three launches each while the pass is thick; from the round the back end names (gfx950: where its survival profile
            // expects at most a wave's worth of paths per segment) ONE launch in which every wave takes its segment through all remaining
            // rounds.  Scenes with null-BSDF skips keep the round-by-round form (their passes stay open for more rounds).
            const uint32_t n_rounds = std::min(fixed_iters + spare_rounds, max_iters);
            const uint32_t tail_from = spare_rounds ? 0xffffffffu : be.tail_round(R.n_paths, n_rounds);
            for (; it < n_rounds && it < tail_from; ++it) round(it);
            if (it < n_rounds) { be.tail(it, n_rounds); it = n_rounds; }
            // output rows touched by sample rows [r0, r1): pixel row = min_y + sample row, +-2
            const int32_t y0 = std::max(rb, g.min_y + r0 - 2), y1 = std::min(re, g.min_y + r1 - 1 + 2 + 1);
            pending[lane].active = true; pending[lane].it = it; pending[lane].n_paths = R.n_paths; pending[lane].pass_no = pass_no;
            pending[lane].y0 = (single_pixel || y1 <= y0) ? 0 : y0; pending[lane].y1 = (single_pixel || y1 <= y0) ? 0 : y1;
            pending[lane].open = true;
            if (!spare_rounds) close_pass(lane); // nothing to ask: the film kernel follows the rounds at once
        }
    }
    st.ms_enqueue = std::chrono::duration<double, std::milli>(clock::now() - t_begin).count();
    for (uint32_t k = 0; k < n_lanes; ++k) finish((pass_no + k) % n_lanes); // oldest pass first: the callbacks keep their order
    be.end(st);
    if (null_skip_overrun) st.error_flags |= PTRS_ERRFLAG_NULL_SKIPS;
    st.bvh_nodes = sc.n_nodes; st.bvh_max_depth = bvh_depth;
    st.ms_total = std::chrono::duration<double, std::milli>(clock::now() - t_begin).count();
    if (stats) *stats = st;
    if (st.error_flags & PTRS_ERRFLAG_SOBOL_DIM) {
        err = strat_dim ? "stratified sampler: a path drew past n_sampled_dimensions (null-BSDF skips lengthen paths beyond max_depth, Q7); the reference would continue from the tile's generator in path order, which a wavefront cannot reproduce"
                        : "sobol sampler can only sample up to 1024 dimensions (sobol.rs:177-183): max_depth is too large for this scene";
        return PTRS_ERR_UNSUPPORTED;
    }
    if (st.error_flags & PTRS_ERRFLAG_NULL_SKIPS) { err = "paths still alive after max_depth + 65 rounds of null-BSDF skips (integrator.rs:434-439)"; return PTRS_ERR_UNSUPPORTED; }
    return PTRS_OK;
}

} // namespace pt
