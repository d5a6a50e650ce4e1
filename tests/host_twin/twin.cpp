// HOST TWIN -- TEST INFRASTRUCTURE ONLY.
// Compiles the product's device code (pathtracer-rs_amd/csrc/pt_*.h: the same stage functions and
// the same orchestration the gfx950 kernels use) for the CPU and runs every "kernel" as a serial
// loop.  Purpose: debug the wavefront pipeline against the oracle in the GPU-less authoring
// container.  It is never loaded by the product (which fails loudly without its HIP library) and
// is not a CPU fallback; only tests/ build and call it.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../pathtracer-rs_amd/csrc/pt_render.h"

using namespace pt;

namespace {

thread_local std::string g_err;
SobolTablesHost g_tables;

struct TwinScene {
    HostScene H;
    DScene sc;
};

struct HostBackend {
    DScene sc; DSampler S; DCamera C; DParams R; DPaths P; DQueues Q;
    std::vector<std::vector<unsigned char>> pool;
    float table[256];
    uint32_t cap = 0, rows = 0;
    uint64_t nodes = 0, tris = 0;

    const uint32_t *sobol_matrices() { return g_tables.matrices.data(); }
    const uint64_t *sobol_vdc(uint32_t row) { return g_tables.vdc.data() + (size_t)row * g_tables.stride; }
    const uint64_t *sobol_vdc_inv(uint32_t row) { return g_tables.vdc_inv.data() + (size_t)row * g_tables.stride; }

    template <class T> T *alloc(size_t n) { pool.emplace_back(n * sizeof(T) + 64); return reinterpret_cast<T *>(pool.back().data()); }

    int begin(const DScene &sc_, const DSampler &S_, const DCamera &C_, uint32_t capacity, uint32_t count_rows, uint32_t, uint32_t, std::string &) {
        sc = sc_; S = S_; C = C_; cap = capacity; rows = count_rows;
        gaussian_filter_table(table);
        P.ray_o = alloc<v4>(cap); P.ray_d = alloc<v4>(cap); P.beta = alloc<v4>(cap); P.L = alloc<v4>(cap); P.st = alloc<u4>(cap); P.hit = alloc<u4>(cap);
        P.pfilm = alloc<v4>(cap); P.nee0 = alloc<v4>(cap); P.nee1 = alloc<v4>(cap); P.nee2 = alloc<u4>(cap); P.sh_o = alloc<v4>(cap); P.sh_d = alloc<v4>(cap);
        P.mis_o = alloc<v4>(cap); P.mis_d = alloc<v4>(cap); P.mis_hit = alloc<u4>(cap); P.sh_res = alloc<uint32_t>(cap);
        Q.ext[0] = alloc<uint32_t>(cap); Q.ext[1] = alloc<uint32_t>(cap);
        for (int k = 0; k < Q_NUM_MAT; ++k) Q.mat[k] = alloc<uint32_t>(cap);
        Q.shadow = alloc<uint32_t>(cap); Q.mis = alloc<uint32_t>(cap); Q.nee = alloc<uint32_t>(cap);
        Q.counts = alloc<uint32_t>((size_t)rows * Q_STRIDE); Q.stats = alloc<unsigned long long>(CNT_NUM);
        return PTRS_OK;
    }
    void pass_begin(const DParams &R_) { R = R_; std::memset(Q.counts, 0, (size_t)rows * Q_STRIDE * 4); }
    uint32_t &cnt(uint32_t it, int q) { return Q.counts[(size_t)it * Q_STRIDE + q]; }
    void generate() {
        for (uint32_t pid = 0; pid < R.n_paths; ++pid) { generate_item(R, S, C, P, pid); Q.ext[0][pid] = pid; }
        cnt(0, Q_EXT) = R.n_paths;
    }
    template <bool ANY> void trace(const uint32_t *queue, uint32_t n, const v4 *ro, const v4 *rd, u4 *hits, uint32_t *occl) {
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t pid = queue[i];
            v4 o = ro[pid], d = rd[pid];
            LocalStack stk; HitRec h; uint32_t nn = 0, nt = 0;
            bool hit = bvh_trace<ANY>(sc, xyz(o), xyz(d), o.w, stk, h, nn, nt);
            nodes += nn; tris += nt;
            if (ANY) occl[pid] = hit ? 1u : 0u;
            else { u4 r; r.x = (uint32_t)h.prim; r.y = f2u(h.b0); r.z = f2u(h.b1); r.w = f2u(h.b2); hits[pid] = r; }
        }
    }
    void trace_extension(uint32_t it) { trace<false>(Q.ext[it & 1], cnt(it, Q_EXT), P.ray_o, P.ray_d, P.hit, nullptr); }
    void trace_shadow(uint32_t it) { trace<true>(Q.shadow, cnt(it, Q_SHADOW), P.sh_o, P.sh_d, nullptr, P.sh_res); }
    void trace_mis(uint32_t it) { trace<false>(Q.mis, cnt(it, Q_MIS), P.mis_o, P.mis_d, P.mis_hit, nullptr); }
    void sort(uint32_t it) {
        const uint32_t *q = Q.ext[it & 1];
        for (uint32_t i = 0, n = cnt(it, Q_EXT); i < n; ++i) {
            int k = sort_item(R, sc, P, q[i]);
            if (k >= 0) Q.mat[k][cnt(it, Q_MAT0 + k)++] = q[i];
        }
    }
    void shade(uint32_t it, int kind) {
        const uint32_t *q = Q.mat[kind];
        uint32_t *next = Q.ext[(it + 1) & 1];
        for (uint32_t i = 0, n = cnt(it, Q_MAT0 + kind); i < n; ++i) {
            uint32_t pid = q[i];
            ShadeResult r = shade_item(R, S, C, sc, P, pid);
            if (r.next) next[cnt(it + 1, Q_EXT)++] = pid;
            if (r.shadow) Q.shadow[cnt(it, Q_SHADOW)++] = pid;
            if (r.mis) Q.mis[cnt(it, Q_MIS)++] = pid;
            if (r.nee) Q.nee[cnt(it, Q_NEE)++] = pid;
        }
    }
    void resolve(uint32_t it) { for (uint32_t i = 0, n = cnt(it, Q_NEE); i < n; ++i) resolve_item(sc, P, Q.nee[i]); }
    uint32_t read_count(uint32_t it, int q) { return cnt(it, q); }
    void read_counts(uint32_t *dst, uint32_t n_rows) { std::memcpy(dst, Q.counts, (size_t)n_rows * Q_STRIDE * 4); }
    void film(v4 *film_px, int32_t y0, int32_t y1) {
        for (int32_t y = y0; y < y1; ++y) for (int32_t x = 0; x < R.W; ++x) film_item(R, S, P, table, film_px, x, y);
    }
    void export_samples(float *out) {
        for (uint32_t pid = 0; pid < R.n_paths; ++pid) {
            PathCoord c = path_coord(R, S, pid);
            size_t o = (((size_t)c.sy * (size_t)R.NX + (size_t)c.sx) * S.spp + c.s) * 3;
            out[o] = P.L[pid].x; out[o + 1] = P.L[pid].y; out[o + 2] = P.L[pid].z;
        }
    }
    void end(PtrsStats &st) { st.nodes_visited = nodes; st.tris_tested = tris; }
};

} // namespace

extern "C" {

const char *twin_last_error(void) { return g_err.c_str(); }
int twin_load_tables(const char *path) { return load_sobol_tables(path, g_tables) ? PTRS_OK : PTRS_ERR_IO; }

int twin_scene_create(const PtrsSceneDesc *d, void **out) {
    auto *s = new TwinScene();
    int rc = build_host_scene(*d, s->H, g_err);
    if (rc != PTRS_OK) { delete s; return rc; }
    HostScene &H = s->H; DScene &sc = s->sc;
    sc.nodes = H.nodes.data(); sc.tris = H.tris.data(); sc.shade = H.shade.data(); sc.mats = H.mats.data(); sc.texs = H.texs.data();
    sc.levels = H.levels.data(); sc.texdata = H.texdata.data(); sc.lights = H.lights.data(); sc.distdata = H.distdata.data(); sc.inf_lights = H.inf_lights.data();
    sc.n_nodes = (uint32_t)H.nodes.size(); sc.n_prims = (uint32_t)H.tris.size(); sc.n_lights = (uint32_t)H.lights.size(); sc.n_inf = (uint32_t)H.inf_lights.size();
    *out = s;
    return PTRS_OK;
}
void twin_scene_destroy(void *s) { delete static_cast<TwinScene *>(s); }

int twin_render(void *sp, const PtrsCamera *cam, const PtrsRenderParams *prm, PtrsFilmPixel *film, float *sample_rgb, PtrsStats *stats) {
    if (!g_tables.ok) { g_err = "tables not loaded"; return PTRS_ERR_INVALID; }
    TwinScene *s = static_cast<TwinScene *>(sp);
    HostBackend be;
    return render_impl(be, s->sc, s->H.kinds_present, s->H.max_depth, *cam, *prm, reinterpret_cast<v4 *>(film), sample_rgb, stats, g_err);
}

int twin_trace_rays(void *sp, uint32_t n, const float *rays, int32_t any_hit, PtrsHit *hits, PtrsStats *stats) {
    TwinScene *s = static_cast<TwinScene *>(sp);
    uint64_t nodes = 0, tris = 0;
    for (uint32_t i = 0; i < n; ++i) {
        LocalStack stk; HitRec h; uint32_t nn = 0, nt = 0;
        f3 o = mk3(rays[7 * i], rays[7 * i + 1], rays[7 * i + 2]), d = mk3(rays[7 * i + 3], rays[7 * i + 4], rays[7 * i + 5]);
        bool hit = any_hit ? bvh_trace<true>(s->sc, o, d, rays[7 * i + 6], stk, h, nn, nt) : bvh_trace<false>(s->sc, o, d, rays[7 * i + 6], stk, h, nn, nt);
        nodes += nn; tris += nt;
        hits[i].prim = any_hit ? (hit ? 0 : -1) : h.prim; hits[i].t = h.t; hits[i].b0 = h.b0; hits[i].b1 = h.b1; hits[i].b2 = h.b2;
    }
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->nodes_visited = nodes; stats->tris_tested = tris; }
    return PTRS_OK;
}

int twin_sobol_samples(const PtrsRenderParams *prm, uint32_t n, const int32_t *px, const int32_t *py, const uint64_t *sample_nums, const uint32_t *dims, float *out, uint64_t *index_out) {
    if (!g_tables.ok) { g_err = "tables not loaded"; return PTRS_ERR_INVALID; }
    SampleGrid g = make_sample_grid(prm->width, prm->height, prm->spp);
    DSampler S;
    S.matrices = g_tables.matrices.data(); S.vdc = g_tables.vdc.data() + (size_t)(g.log2_res - 1) * g_tables.stride; S.vdc_inv = g_tables.vdc_inv.data() + (size_t)(g.log2_res - 1) * g_tables.stride;
    S.log2_res = g.log2_res; S.resolution = g.resolution; S.min_x = g.min_x; S.min_y = g.min_y; S.spp = g.spp;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t idx = sobol_index(S, sample_nums[i], (uint32_t)(px[i] - g.min_x), (uint32_t)(py[i] - g.min_y));
        if (index_out) index_out[i] = idx;
        out[i] = sample_dimension(S, idx, dims[i], pixel_scramble(px[i], py[i]), px[i], py[i]);
    }
    return PTRS_OK;
}

} // extern "C"
