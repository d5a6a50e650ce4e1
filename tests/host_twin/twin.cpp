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
std::vector<uint32_t> g_bytetab;
bool g_use_bytetab = true;
bool g_force_full = false;

struct TwinScene {
    HostScene H;
    DScene sc;
};

// traversal stack with the capacity the GPU kernel would get (16 / 32 / 64 LDS entries) and an
// overflow flag, so that a stack misuse shows up on the CPU instead of corrupting LDS on the GPU
struct CheckedStack {
    uint32_t s[128]; float te[128]; int n = 0; int cap = 128; bool *overflow = nullptr;
    void push(uint32_t v, float t) { if (n >= cap) { if (overflow) *overflow = true; return; } s[n] = v; te[n] = t; ++n; }
    void pop(uint32_t &v, float &t) { --n; v = s[n]; t = te[n]; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
};

struct HostBackend {
    bool overflow = false, err_dim = false; int stack_cap = 128;
    CheckedStack make_stack() { CheckedStack k; k.cap = stack_cap; k.overflow = &overflow; return k; }
    DScene sc; DSampler S; DCamera C; DParams R; DPaths P; DQueues Q;
    std::vector<std::vector<unsigned char>> pool;
    float table[256];
    uint32_t cap = 0, rows = 0;
    int feat = FEAT_FULL, feat_trace = FEAT_FULL;
    uint64_t nodes = 0, tris = 0;

    const uint32_t *sobol_matrices() { return g_tables.matrices.data(); }
    const uint32_t *sobol_bytetab() { return g_use_bytetab ? g_bytetab.data() : nullptr; }
    const uint32_t *sobol_nibtab() { return nullptr; }
    std::vector<float> strat1, strat2;
    int strat_tables(int32_t NX, int32_t NY, uint32_t dim_ps, uint32_t n_dims, const float **t1, const float **t2, std::string &) {
        const size_t n = (size_t)NX * (size_t)NY * n_dims * dim_ps * dim_ps;
        strat1.assign(n, 0.0f); strat2.assign(n * 2, 0.0f);
        const int32_t ntx = (NX + 15) / 16, nty = (NY + 15) / 16;
        for (int32_t ty = 0; ty < nty; ++ty)
            for (int32_t tx = 0; tx < ntx; ++tx)
                stratified_tile_tables((uint64_t)(ty * ntx + tx), tx * 16, std::min(tx * 16 + 16, NX), ty * 16, std::min(ty * 16 + 16, NY), NX, dim_ps, n_dims, strat1.data(), strat2.data());
        *t1 = strat1.data(); *t2 = strat2.data();
        return PTRS_OK;
    }
    const uint64_t *sobol_vdc(uint32_t row) { return g_tables.vdc.data() + (size_t)row * g_tables.stride; }
    const uint64_t *sobol_vdc_inv(uint32_t row) { return g_tables.vdc_inv.data() + (size_t)row * g_tables.stride; }

    template <class T> T *alloc(size_t n) { pool.emplace_back(n * sizeof(T) + 64); return reinterpret_cast<T *>(pool.back().data()); }

    int begin(const DScene &sc_, const DSampler &S_, const DCamera &C_, uint32_t capacity, uint32_t count_rows, uint32_t bvh_depth, uint32_t, int feat_, int feat_trace_, std::string &) {
        feat_trace = g_force_full ? FEAT_FULL : feat_trace_;
        (void)bvh_depth;
        sc = sc_; S = S_; C = C_; cap = capacity; rows = count_rows; feat = g_force_full ? FEAT_FULL : feat_;
        gaussian_filter_table(table);
        // (one queue segment: a path's position in a queue is the index of its entry)
        for (int b = 0; b < 2; ++b) { P.ray_o[b] = alloc<v4>(cap); P.ray_d[b] = alloc<v4>(cap); P.beta[b] = alloc<v4>(cap); }
        P.L = alloc<v4>(cap); P.st = alloc<u4>(cap); P.hit = alloc<u4>(cap); P.pre0 = P.pre1 = nullptr;
        P.pfilm = alloc<f2a>(cap); P.nee0 = alloc<v4>(cap); P.nee1 = alloc<v4>(cap); P.nee2 = alloc<u4>(cap); P.sh_o = alloc<v4>(cap); P.sh_d = alloc<v4>(cap);
        P.mis_o = alloc<v4>(cap); P.mis_d = alloc<v4>(cap); P.nhit = alloc<u4>(cap); P.pre_z = alloc<float>(cap);
        Q.ext[0] = alloc<uint32_t>(cap); Q.ext[1] = alloc<uint32_t>(cap);
        for (int k = 0; k < Q_NUM_MAT; ++k) Q.mat[k] = alloc<MatEntry>(cap);
        Q.nee = alloc<uint32_t>(cap);
        Q.counts = alloc<uint32_t>((size_t)rows * Q_STRIDE); Q.stats = alloc<unsigned long long>(CNT_NUM);
        return PTRS_OK;
    }
    uint32_t lanes(uint64_t = 0, const bool * = nullptr, bool = false, uint32_t = 0) const { return 1; }
    uint64_t auto_capacity(uint32_t, const bool *) const { return 1ull << 27; }
    void select(uint32_t) {}
    void pass_begin(const DParams &R_) { R = R_; std::memset(Q.counts, 0, (size_t)rows * Q_STRIDE * 4); }
    uint32_t &cnt(uint32_t it, int q) { return Q.counts[(size_t)it * Q_STRIDE + q]; }
    void generate() {
        for (uint32_t pid = 0; pid < R.n_paths; ++pid) { generate_item(R, S, C, P, pid, pid); Q.ext[0][pid] = pid; }
        cnt(0, Q_EXT) = R.n_paths;
    }
    template <int FEAT> void extend_t(uint32_t it) {
        const uint32_t *q = Q.ext[it & 1], par = it & 1u;
        for (uint32_t i = 0, n = cnt(it, Q_EXT); i < n; ++i) {
            const uint32_t pid = q[i];
            const v4 o = P.ray_o[par][i], d = P.ray_d[par][i];
            CheckedStack stk = make_stack(); HitRec h; uint32_t nn = 0, nt = 0;
            bvh_trace_any_form<false, (FEAT & FEAT_ALPHA) != 0>(sc, xyz(o), xyz(d), PT_INF, stk, h, nn, nt);
            nodes += nn; tris += nt;
            u4 r; r.x = hit_pack(h.prim, h.flags); r.y = f2u(h.b0); r.z = f2u(h.b1); r.w = f2u(h.b2); P.hit[i] = r;
            const int k = extension_epilogue<FEAT>(R, sc, P, par, pid, i, h);
            if (k >= 0) { MatEntry me; me.e = i; me.pid = pid; Q.mat[k][cnt(it, Q_MAT0 + k)++] = me; }
        }
    }
    void extend(uint32_t it) { if (feat_trace != FEAT_SIMPLE) extend_t<FEAT_FULL>(it); else extend_t<FEAT_SIMPLE>(it); } // (FEAT_IMG_ENV: the full set gives the same results)
    void shade(uint32_t it, int kind) {
        const MatEntry *q = Q.mat[kind];
        uint32_t *next = Q.ext[(it + 1) & 1];
        const uint32_t par = it & 1u;
        for (uint32_t i = 0, n = cnt(it, Q_MAT0 + kind); i < n; ++i) {
            const uint32_t pid = q[i].pid, e = q[i].e;
            const ShadeResult r = feat == FEAT_SIMPLE ? shade_dispatch<FEAT_SIMPLE>(kind, R, S, C, sc, P, par, pid, e) : feat == FEAT_IMG ? shade_dispatch<FEAT_IMG>(kind, R, S, C, sc, P, par, pid, e)
                                : feat == FEAT_IMG_ENV ? shade_dispatch<FEAT_IMG_ENV>(kind, R, S, C, sc, P, par, pid, e) : shade_dispatch<FEAT_FULL>(kind, R, S, C, sc, P, par, pid, e);
            err_dim = err_dim || r.err_dim;
            store_shade_out(P, par ^ 1u, r, cnt(it + 1, Q_EXT), cnt(it, Q_NEE)); // (the positions the two pushes below hand out)
            if (r.next) next[cnt(it + 1, Q_EXT)++] = pid;
            if (r.shadow) cnt(it, Q_SHADOW)++;
            if (r.mis) cnt(it, Q_MIS)++;
            if (r.nee) Q.nee[cnt(it, Q_NEE)++] = r.nee_entry(pid);
        }
    }
    void connect(uint32_t it) {
        for (uint32_t i = 0, n = cnt(it, Q_NEE); i < n; ++i) {
            CheckedStack stk = make_stack(); uint32_t nn = 0, nt = 0;
            const GeomGlobal G = geom_global(sc);
            if (sc.n_nodes4) { if (feat_trace != FEAT_SIMPLE) connect_item<FEAT_FULL, true>(sc, G, P, Q.nee[i], i, stk, nn, nt); else connect_item<FEAT_SIMPLE, true>(sc, G, P, Q.nee[i], i, stk, nn, nt); }
            else { if (feat_trace != FEAT_SIMPLE) connect_item<FEAT_FULL, false>(sc, G, P, Q.nee[i], i, stk, nn, nt); else connect_item<FEAT_SIMPLE, false>(sc, G, P, Q.nee[i], i, stk, nn, nt); }
            nodes += nn; tris += nt;
        }
    }
    uint32_t read_count(uint32_t it, int q) { return cnt(it, q); }
    void read_counts(uint32_t *dst, uint32_t n_rows) { std::memcpy(dst, Q.counts, (size_t)n_rows * Q_STRIDE * 4); }
    void film(v4 *film_px, int32_t y0, int32_t y1) {
        for (int32_t y = y0; y < y1; ++y) for (int32_t x = 0; x < R.W; ++x) film_item(R, S, P, table, film_px, x, y);
    }
    void publish_rows(v4 *, int32_t, int32_t) {}
    void dump_rays(uint32_t, const RayDump &) {}
    void presample(uint32_t) {}
    uint32_t tail_round(uint32_t, uint32_t) const { return 0xffffffffu; } // (the fused tail is a launch-shape matter of the gfx950 back end)
    void tail(uint32_t, uint32_t) {}
    void learn(const uint32_t *, uint32_t, uint32_t) {}
    void export_samples(float *out) {
        for (uint32_t pid = 0; pid < R.n_paths; ++pid) {
            PathCoord c = path_coord(R, S, pid);
            size_t o = R.pixel_mode ? (size_t)c.s * 3 : (((size_t)c.sy * (size_t)R.NX + (size_t)c.sx) * S.spp + c.s) * 3;
            out[o] = P.L[pid].x; out[o + 1] = P.L[pid].y; out[o + 2] = P.L[pid].z;
        }
    }
    void end(PtrsStats &st) { st.nodes_visited = nodes; st.tris_tested = tris; if (overflow) st.kernel_launches = 0xdeadull; if (err_dim) st.error_flags |= PTRS_ERRFLAG_SOBOL_DIM; }
};

} // namespace

extern "C" {

const char *twin_last_error(void) { return g_err.c_str(); }
int twin_load_tables(const char *path) {
    if (!load_sobol_tables(path, g_tables)) return PTRS_ERR_IO;
    g_tables.matrices.resize(g_tables.matrices.size() + 16 * 52, 0u); // head-room for a path that overruns the 1024 dimensions (it raises PTRS_ERRFLAG_SOBOL_DIM)
    build_sobol_bytetab(g_tables, g_bytetab);
    return PTRS_OK;
}
void twin_use_bytetab(int on) { g_use_bytetab = on != 0; }
void twin_force_full_features(int on) { g_force_full = on != 0; }

int twin_scene_create(const PtrsSceneDesc *d, void **out) {
    auto *s = new TwinScene();
    int rc = build_host_scene(*d, s->H, g_err);
    if (rc != PTRS_OK) { delete s; return rc; }
    HostScene &H = s->H; DScene &sc = s->sc;
    sc.nodes2 = H.nodes2.data(); sc.n_nodes2 = (uint32_t)H.nodes2.size(); sc.nodes4 = H.nodes4.data(); sc.n_nodes4 = (uint32_t)H.nodes4.size();
    sc.nodes = H.nodes.data(); sc.tris = H.tris.data(); sc.shade = H.shade.data(); sc.mats = H.mats.data(); sc.texs = H.texs.data();
    sc.levels = H.levels.data(); sc.texdata = H.texdata.data(); sc.lights = H.lights.data(); sc.distdata = H.distdata.data(); sc.inf_lights = H.inf_lights.data();
    sc.n_nodes = (uint32_t)H.nodes.size(); sc.n_prims = (uint32_t)H.tris.size(); sc.n_lights = (uint32_t)H.lights.size(); sc.n_inf = (uint32_t)H.inf_lights.size();
    *out = s;
    return PTRS_OK;
}
void twin_scene_destroy(void *s) { delete static_cast<TwinScene *>(s); }
void twin_scene_info(void *s, uint32_t *max_depth, uint32_t *stack_bound, uint32_t *n_nodes2) { auto *t = static_cast<TwinScene *>(s); *max_depth = t->H.max_depth; *stack_bound = t->H.stack_bound; *n_nodes2 = (uint32_t)(t->H.nodes2.size() + t->H.nodes4.size()); }

int twin_render(void *sp, const PtrsCamera *cam, const PtrsRenderParams *prm, PtrsFilmPixel *film, float *sample_rgb, PtrsStats *stats) {
    if (!g_tables.ok) { g_err = "tables not loaded"; return PTRS_ERR_INVALID; }
    TwinScene *s = static_cast<TwinScene *>(sp);
    HostBackend be;
    be.stack_cap = (int)std::min<uint32_t>(s->H.stack_bound, 128u); // exactly the host's bound: the GPU sizes LDS + spill from it
    int rc = render_impl(be, s->sc, s->H, s->H.max_depth, *cam, *prm, reinterpret_cast<v4 *>(film), sample_rgb, stats, g_err);
    if (rc == PTRS_OK && be.overflow) { g_err = "traversal stack overflow (would corrupt LDS on the GPU)"; return PTRS_ERR_INVALID; }
    return rc;
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
    S.matrices = g_tables.matrices.data(); S.bytetab = g_use_bytetab ? g_bytetab.data() : nullptr; S.nibtab = nullptr; S.vdc = g_tables.vdc.data() + (size_t)(g.log2_res - 1) * g_tables.stride; S.vdc_inv = g_tables.vdc_inv.data() + (size_t)(g.log2_res - 1) * g_tables.stride;
    S.log2_res = g.log2_res; S.resolution = g.resolution; S.min_x = g.min_x; S.min_y = g.min_y; S.spp = g.spp;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t idx = sobol_index(S, sample_nums[i], (uint32_t)(px[i] - g.min_x), (uint32_t)(py[i] - g.min_y));
        if (index_out) index_out[i] = idx;
        out[i] = sample_dimension(S, idx, dims[i], pixel_scramble(px[i], py[i]), px[i], py[i]);
    }
    return PTRS_OK;
}

} // extern "C"
