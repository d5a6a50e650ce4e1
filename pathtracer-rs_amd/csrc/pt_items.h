// pt_items.h -- the per-path work of every wavefront stage, as plain functions of a path slot.
//
// The pipeline restates PathIntegrator::li (src/pathtracer/integrator.rs:392-503) as a loop over
// stages; one loop iteration of the reference = one pass through
//     extend (trace + epilogue/bucketing) -> shade[material] -> connect (shadow + MIS queries, resolve)
// with path state in HBM between stages.  The functions below do the arithmetic; the HIP kernels
// in kernels.hip (and, for CPU debugging only, tests/host_twin) wrap them with the queue logic.
//   generate_item : integrator.rs:571-577 + sobol.rs:81-120 + pathtracer/mod.rs:59-81 + ray.rs:30-35
//   extension_epilogue : integrator.rs:418-431 (emission, miss, depth cut) + material bucket
//   shade_item<MAT>    : integrator.rs:433-499, uniform_sample_one_light 192-217, estimate_direct
//                        23-135 up to (not including) its two scene queries
//   connect_item       : those two queries (shadow any-hit, MIS closest-hit) and estimate_direct's
//                        use of their results (66-78, 121-134), `l += ld` (444-446)
//   film_item     : FilmTile::add_sample (film.rs:60-106) turned into a per-pixel gather
#pragma once
#include "pt_bvh.h"
#include "pt_light.h"
#include "pt_sobol.h"
#include "pt_stratified.h"

namespace pt {

struct CamRay { f3 o, d, rx_d, ry_d; };

// Camera::generate_ray_differential (pathtracer/mod.rs:59-81) followed by
// scale_differentials(1/sqrt(spp)) (ray.rs:30-35).  rx_origin == ry_origin == o.
PT_HD CamRay camera_ray(const DCamera &C, f2 pf, float diff_scale) {
    const float *m = C.r2s;
    f3 s = mk3(m[0] * pf.x + m[1] * pf.y + m[2] * 0.0f + m[3], m[4] * pf.x + m[5] * pf.y + m[6] * 0.0f + m[7], m[8] * pf.x + m[9] * pf.y + m[10] * 0.0f + m[11]);
    float inv_denom = C.m23 / (s.z + C.m22); // Perspective3::unproject_point
    f3 pc = mk3(s.x * inv_denom / C.m00, s.y * inv_denom / C.m11, -inv_denom);
    CamRay r;
    r.o = quat_rotate(C.rot, splat3(0.0f)) + ld3(C.trans);
    f3 wd = quat_rotate(C.rot, pc);
    r.d = normalize(wd);
    f3 rxd = normalize(quat_rotate(C.rot, pc + ld3(C.dxc)));
    f3 ryd = normalize(quat_rotate(C.rot, pc + ld3(C.dyc)));
    r.rx_d = r.d + (rxd - r.d) * diff_scale;
    r.ry_d = r.d + (ryd - r.d) * diff_scale;
    return r;
}

struct PathCoord { int32_t sx, sy, px, py; uint32_t s; };
PT_HD PathCoord path_coord(const DParams &R, const DSampler &S, uint32_t pid) {
    if (R.pixel_mode) { PathCoord c; c.s = R.s0 + pid; c.sx = R.pix_sx; c.sy = R.pix_sy; c.px = S.min_x + c.sx; c.py = S.min_y + c.sy; return c; }
    const uint32_t npix = (uint32_t)(R.row1 - R.row0) * (uint32_t)R.NX;
    uint32_t pl = pid % npix;
    PathCoord c;
    c.s = R.s0 + pid / npix;
    c.sx = (int32_t)(pl % (uint32_t)R.NX);
    c.sy = R.row0 + (int32_t)(pl / (uint32_t)R.NX);
    c.px = S.min_x + c.sx; c.py = S.min_y + c.sy;
    return c;
}

PT_HD uint32_t strat_pack(uint32_t d1, uint32_t d2) { return (d1 & 63u) | ((d2 & 63u) << 6); } // the stratified sampler's 1-D / 2-D dimension counters in the state word's dimension field

PT_HD void generate_item(const DParams &R, const DSampler &S, const DCamera &C, const DPaths &P, uint32_t pid, uint32_t e) { // e: the path's position in round 0's extension queue
    PathCoord c = path_coord(R, S, pid);
    f2 u; u4 st;
    if (S.kind == PTRS_SAMPLER_STRATIFIED) { // get_camera_sample = the first 2-D dimension of the pixel's table (mod.rs:156-160)
        const uint32_t pixel = (uint32_t)c.sy * (uint32_t)R.NX + (uint32_t)c.sx;
        const float *t = S.strat2 + (((size_t)pixel * S.strat_dims + 0u) * S.spp + c.s) * 2u;
        u = mk2(t[0], t[1]);
        st.x = pixel; st.y = c.s; st.z = strat_pack(0u, 1u) | ST_HAS_DIFF; st.w = 0;
    } else {
        SamplerState ss;
        ss.index = sobol_index(S, (uint64_t)c.s, (uint32_t)c.sx, (uint32_t)c.sy);
        ss.dim = 0; ss.scramble = pixel_scramble(c.px, c.py); ss.px = c.px; ss.py = c.py;
        u = get_2d(S, ss);
        st.x = (uint32_t)ss.index; st.y = (uint32_t)(ss.index >> 32); st.z = ss.dim | ST_HAS_DIFF; st.w = ss.scramble; // the pixel's scramble travels with the path
    }
    f2 pf = mk2((float)c.px + u.x, (float)c.py + u.y);
    CamRay r = camera_ray(C, pf, R.inv_sqrt_spp);
    pslot(P.ray_o[0], e) = mkv4(r.o, PT_INF);
    pslot(P.ray_d[0], e) = mkv4(r.d, u2f(st.z)); // the path's state word (dimension counter | flags | bounces) travels with the ray direction: a vertex
                                          // that continues rewrites ray_d anyway, and `st` keeps only what never changes (Sobol' index, scramble)
    pslot(P.beta[0], e) = mkv4(splat3(1.0f), 1.0f);
    pslot(P.L, pid) = mkv4(splat3(0.0f), 0.0f);
    pslot(P.st, pid) = st;
    { f2a q; q.x = pf.x; q.y = pf.y; pslot(P.pfilm, pid) = q; }
}

// t_max of every shadow ray: spawn_ray_to_it's 1 - 0.0001 (interaction.rs:50-60, Q13)
#define PT_SHADOW_TMAX (1.0f - 0.0001f)

PT_HD int32_t st_bounces(uint32_t z) { return (int32_t)(int16_t)(z >> ST_BOUNCE_SHIFT); }
PT_HD uint32_t st_pack(uint32_t dim, uint32_t flags, int32_t bounces) { return (dim & ST_DIM_MASK) | flags | ((uint32_t)(uint16_t)(int16_t)bounces << ST_BOUNCE_SHIFT); }


// Epilogue of the extension trace (integrator.rs:418-431): emission at the first vertex / after a
// specular bounce, environment radiance for escaped rays, the depth cut -- then the material bucket.
// Returns the bucket (0..5) when the path goes on to shading, -1 when it ends here.
template <int FEAT>
PT_HD int extension_epilogue(const DParams &R, const DScene &sc, const DPaths &P, uint32_t par, uint32_t pid, uint32_t e, const HitRec &h) { // par: the round's parity; e: the path's position in the round's extension queue
    const v4 rdv = pslot(P.ray_d[par], e);
    const uint32_t stz = f2u(rdv.w);
    const int32_t prim = h.prim;
    const int32_t bounces = st_bounces(stz);
    if (bounces == 0 || (stz & ST_SPECULAR)) {
        if (prim >= 0) {
            if (h.flags & TRI_IS_LIGHT) {
                const TriRegs T = load_tri_regs(sc.shade + prim);
                f3 d = xyz(rdv);
                Surface s = tri_surface(T, prim, h.b0, h.b1, h.b2, -d);
                f3 le = surface_le<FEAT>(sc, T, s, -d);
                v4 Lv = pslot(P.L, pid);
                f3 L = xyz(Lv) + xyz(pslot(P.beta[par], e)) * le;
                pslot(P.L, pid) = mkv4(L, Lv.w);
            }
        } else if ((FEAT & FEAT_INFINITE) && sc.n_inf > 0) {
            f3 d = xyz(rdv);
            v4 Lv = pslot(P.L, pid);
            f3 L = xyz(Lv), beta = xyz(pslot(P.beta[par], e));
            for (uint32_t i = 0; i < sc.n_inf; ++i) L = L + beta * light_le<FEAT>(sc, sc.lights[sc.inf_lights[i]], d);
            pslot(P.L, pid) = mkv4(L, Lv.w);
        }
    }
    if (prim < 0 || bounces >= R.max_depth) return -1;
    return (int)((h.flags >> TRI_BUCKET_SHIFT) & 7u); // material bucket, carried by the leaf triangle record
}

// The extension hit's first word: triangle id in bits 0..26 (REF_FIRST_MASK bounds the triangle count), the leaf record's
// material bucket in bits 27..29 and its is-light flag in bit 30, so the epilogue kernel needs no gather from the triangle
// table; 0xffffffff = no hit.
PT_HD uint32_t hit_pack(int32_t prim, uint32_t flags) {
    return prim < 0 ? 0xffffffffu : ((uint32_t)prim | (((flags >> TRI_BUCKET_SHIFT) & 7u) << 27) | ((flags & TRI_IS_LIGHT) ? 1u << 30 : 0u));
}
PT_HD int32_t hit_prim(uint32_t x) { return x == 0xffffffffu ? -1 : (int32_t)(x & REF_FIRST_MASK); }
PT_HD uint32_t hit_flags(uint32_t x) { return x == 0xffffffffu ? 0u : ((((x >> 27) & 7u) << TRI_BUCKET_SHIFT) | (((x >> 30) & 1u) ? (uint32_t)TRI_IS_LIGHT : 0u)); }

// In-kernel stamps (diagnostic builds only, -DPTRS_STAMPS, tools/ablate.sh): wave-clock deltas between the phases of
// shade_item, summed per phase into Q.stats[CNT_STAMP0 + k].  `dep` names values that must exist before the stamp is
// taken, so the wait for their loads falls into the phase before it.  The product build compiles none of this.
#if defined(PTRS_STAMPS) && defined(__HIPCC__)
#define PT_STAMP_PARAMS , unsigned long long *stamp_acc, unsigned long long &stamp_last
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_STAMP(k, dep) { unsigned long long t_; const uint32_t d_ = (dep); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(d_) : "memory"); stamp_acc[k] += t_ - stamp_last; stamp_last = t_; }
#else
#define PT_STAMP(k, dep)
#endif
#else
#define PT_STAMP_PARAMS
#if defined(PTRS_SCHED_BARRIERS) && defined(__HIP_DEVICE_COMPILE__)
#define PT_STAMP(k, dep) __builtin_amdgcn_sched_barrier(0); // A/B build: the phases of shade_item are not interleaved by the scheduler (shorter live ranges, fewer spills?)
#else
#define PT_STAMP(k, dep)
#endif
#endif

// The draws of one shading vertex, for either sampler.  `field` is the dimension field of the path's state word: the Sobol'
// dimension counter, or the stratified sampler's two counters (1-D in bits 0-5, 2-D in bits 6-11; mod.rs:100-101).
struct VertexDraws { float u_nee[5], u_tail[3]; uint32_t after_cont, after_rr; bool err_cont, err_rr; };
template <class CTX>
PT_HD void draw_vertex(const CTX &X, const DSampler &S, const u4 &stv, bool nee, VertexDraws &D) {
    const uint32_t field = stv.z & ST_DIM_MASK;
    for (int k = 0; k < 5; ++k) D.u_nee[k] = 0.0f;
    if (S.kind == PTRS_SAMPLER_STRATIFIED) {
        // get_2d / get_1d read samples_2d[current_2d_dimension][sample] / samples_1d[...] and advance their counter (mod.rs:129-154);
        // past n_sampled_dimensions they would draw from the generator: flagged, the render is refused
        uint32_t d1 = field & 63u, d2 = (field >> 6) & 63u;
        const uint32_t nd = S.strat_dims, pixel = stv.x, sidx = stv.y;
        auto t1 = [&](uint32_t d) { return S.strat1[((size_t)pixel * nd + (d < nd ? d : nd - 1u)) * S.spp + sidx]; };
        auto t2 = [&](uint32_t d, int c) { return S.strat2[(((size_t)pixel * nd + (d < nd ? d : nd - 1u)) * S.spp + sidx) * 2u + (uint32_t)c]; };
        bool err = false;
        if (nee) {
            D.u_nee[0] = t2(d2, 0); D.u_nee[1] = t2(d2, 1); D.u_nee[2] = t2(d2 + 1u, 0); D.u_nee[3] = t2(d2 + 1u, 1); D.u_nee[4] = t1(d1);
            err = d2 + 1u >= nd || d1 >= nd;
            d2 += 2u; d1 += 1u;
        }
        D.u_tail[0] = t2(d2, 0); D.u_tail[1] = t2(d2, 1);
        err = err || d2 >= nd;
        d2 += 1u;
        D.u_tail[2] = t1(d1);
        D.after_cont = strat_pack(d1, d2); D.err_cont = err;
        D.after_rr = strat_pack(d1 + 1u, d2); D.err_rr = err || d1 >= nd;
        return;
    }
    const uint64_t index = (uint64_t)stv.x | ((uint64_t)stv.y << 32);
    const VertexDims V = vertex_dims(field, nee);
    const uint32_t dt[3] = {V.cont[0], V.cont[1], V.rr};
    if (nee) {
        const uint32_t dn[8] = {V.nee[0], V.nee[1], V.nee[2], V.nee[3], V.nee[4], dt[0], dt[1], dt[2]};
        float u8[8];
        X.template sobol<8>(S, index, dn, stv.w, u8);
        for (int k = 0; k < 5; ++k) D.u_nee[k] = u8[k];
        D.u_tail[0] = u8[5]; D.u_tail[1] = u8[6]; D.u_tail[2] = u8[7];
    } else X.template sobol<3>(S, index, dt, stv.w, D.u_tail);
    D.after_cont = V.after_cont; D.err_cont = V.after_cont > 1024u; // a dimension >= 1024 was drawn: the reference panics (sobol.rs:177-183)
    D.after_rr = V.after_cont + 1u; D.err_rr = V.after_cont + 1u > 1024u;
}

// Where shade_item finds its read-only tables.  This one reads everything from global memory (host twin, and the base of
// the gfx950 context in ptrs_hip.hip, which serves the light records, small scenes' triangle records and the Sobol' tables
// of the current round out of LDS: the shade kernels are bound by the number of vector-memory instructions in flight, not
// by bytes).
struct ShadeCtx {
    static constexpr bool inf_fallback = true; // light_sample_li may be asked for an InfiniteAreaLight sample
    PT_MEM TriRegs tri(const DScene &sc, int32_t prim, bool want_dp) const { return load_tri_regs(sc.shade + prim, want_dp); }
    PT_MEM void light(const DScene &sc, uint32_t li, DLight &out) const { out = sc.lights[li]; }
    PT_MEM const InfMarginal *inf_marginal(uint32_t) const { return nullptr; }
    PT_MEM bool presampled(uint32_t) const { return false; } // (the gfx950 pipeline evaluates the environment light's samples of a round ahead of its shade kernels: k_env_presample)
    template <int N> PT_MEM void sobol(const DSampler &S, uint64_t index, const uint32_t (&dim)[N], uint32_t scramble, float (&out)[N]) const { sobol_batch<N>(S, index, dim, scramble, out); }
    PT_MEM void before_stores() const {}
};

// What a shading vertex leaves behind.  shade_item computes it and stores nothing: WHERE it goes is the caller's business -- the
// continuing ray at the position the path gets in the next round's extension queue, the NEE record at the position of its entry in
// the NEE queue (store_shade_out below, once the caller has handed out the positions).
struct ShadeResult {
    bool next; bool nee; bool shadow; bool mis; bool err_dim; // err_dim: a Sobol dimension >= 1024 was drawn (the reference panics, sobol.rs:177-183)
    v4 ro, rd, beta;                     // next: the continuing ray (rd.w: the state word) and the throughput
    v4 sh_o, sh_d, mis_o, mis_d, nee0, nee1; u4 nee2; float pre_z; // nee: the record (shadow: sh_*; mis: mis_*, nee0-2; neither mis: NEE_PRE, its contribution in sh_d.w, sh_o.w, pre_z)
    PT_MEM uint32_t nee_entry(uint32_t pid) const { return pid | (shadow ? (uint32_t)NEE_Q_SHADOW : 0u) | (mis ? (uint32_t)NEE_Q_MIS : (uint32_t)NEE_Q_PRE); }
};
// e_next: the path's position in the next round's extension queue (used if r.next), f: the position of its NEE-queue entry (if r.nee)
PT_HD void store_shade_out(const DPaths &P, uint32_t par_next, const ShadeResult &r, uint32_t e_next, uint32_t f) {
    if (r.shadow) { pstore(P.sh_o, f, r.sh_o); pstore(P.sh_d, f, r.sh_d); }
    if (r.mis) { pstore(P.mis_o, f, r.mis_o); pstore(P.mis_d, f, r.mis_d); pstore(P.nee0, f, r.nee0); pstore(P.nee1, f, r.nee1); pstore(P.nee2, f, r.nee2); } // (a record without a MIS ray is NEE_PRE: nothing in nee0 / nee1 / nee2)
    else if (r.nee) pstore(P.pre_z, f, r.pre_z);
    if (r.next) { pstore(P.ray_o[par_next], e_next, r.ro); pstore(P.ray_d[par_next], e_next, r.rd); pstore(P.beta[par_next], e_next, r.beta); }
}

// What a shading vertex reads of its path: five 16-byte vectors out of HBM (nothing else of a path is cache-resident: a
// pass holds tens of GB of path state).  The gfx950 shade kernel fetches the NEXT item's PathIn while it shades the
// current one (k_shade), which hides the one HBM round trip of the stage.
struct PathIn { v4 ro, rd, beta; u4 st, hit; v4 pre0, pre1; }; // pre0 / pre1: the vertex's presampled environment-light sample (wi, pdf | Li, valid), where the context says so
PT_HD PathIn load_path_in(const DPaths &P, uint32_t par, uint32_t pid, uint32_t e) { PathIn p; p.ro = pslot(P.ray_o[par], e); p.rd = pslot(P.ray_d[par], e); p.beta = pslot(P.beta[par], e); p.st = pslot(P.st, pid); p.hit = pslot(P.hit, e); p.pre0 = p.pre1 = mkv4(splat3(0.0f), 0.0f); return p; }

template <int MAT, int FEAT, class CTX = ShadeCtx>
PT_HD ShadeResult shade_item(const DParams &R, const DSampler &S, const DCamera &C, const DScene &sc, const DPaths &P, uint32_t pid, const PathIn &in, const CTX &X PT_STAMP_PARAMS) {
    ShadeResult out; out.next = false; out.nee = false; out.shadow = false; out.mis = false; out.err_dim = false; out.pre_z = 0.0f;
    // ---- memory round trip 1: the path's state (already requested by the caller) ------------------------------------
    const v4 rov = in.ro, rdv = in.rd;
    v4 bv = in.beta;
    u4 stv = in.st; stv.z = f2u(rdv.w); // the state word arrives in ray_d.w (generate_item)
    const u4 h = in.hit;
    const f3 ro = xyz(rov), rd = xyz(rdv);
    f3 beta = xyz(bv);
    float eta_scale = bv.w;
    const int32_t prim = hit_prim(h.x);
    int32_t bounces = st_bounces(stv.z);
    const uint32_t dim0 = stv.z & ST_DIM_MASK;
    PT_STAMP(0, h.x + stv.z + f2u(rov.x) + f2u(rdv.x) + f2u(bv.x))
    // ---- round trip 2: the triangle's shading record and ALL Sobol' table entries this vertex can need -----------
    // Whether the vertex does next-event estimation is a property of the material kind, except for a Substrate without
    // lobes (Q20) and a glass that yields no BSDF (Q17): the draws below assume the kind's usual answer; the rare other
    // case draws again further down (same function of (index, dimension), so the same values).
    constexpr bool NEE_KIND = MAT != PTRS_MAT_MIRROR && MAT != PTRS_MAT_GLASS;
    const bool nee_guess = NEE_KIND && sc.n_lights > 0;
    const TriRegs T = X.tri(sc, prim, (FEAT & FEAT_IMAGE) != 0);
    VertexDraws D;
    draw_vertex(X, S, stv, nee_guess, D);
    float (&u_nee)[5] = D.u_nee; float (&u_tail)[3] = D.u_tail;
    PT_STAMP(1, f2u(T.ng.x) + f2u(T.p0.x) + f2u(T.n0.x) + f2u(T.uv0.x) + f2u(u_tail[2]) + f2u(u_nee[4]) + T.flags)
    // ---- round trip 3: the light record (its index is the fifth draw) and the material record ---------------------
    uint32_t li = 0;
    if (nee_guess) {
        const float fl = floor_(u_nee[4] * (float)sc.n_lights);
        li = fl > 0.0f ? (uint32_t)fl : 0u;
        if (li > sc.n_lights - 1u) li = sc.n_lights - 1u;
    }
    PT_STAMP(2, li)
    const f3 wo = -rd;
    Surface s = tri_surface(T, prim, u2f(h.y), u2f(h.z), u2f(h.w), wo);
    // Only the camera ray carries differentials (Q9) and only image-texture lookups read them
    // (texture.rs:185-191, 430-445), so without image textures they are dead values.
    if ((FEAT & FEAT_IMAGE) && (stv.z & ST_HAS_DIFF)) {
        const f2a pf = pslot(P.pfilm, pid);
        CamRay cr = camera_ray(C, mk2(pf.x, pf.y), R.inv_sqrt_spp);
        surface_differentials(s, ro, cr.rx_d, ro, cr.ry_d);
    }
    // Everything this vertex writes is collected in `out` and stored by the caller (store_shade_out), behind X.before_stores(): the
    // gfx950 kernel waits there for the NEXT item's prefetched state (vmcnt counts loads and stores alike, so waiting
    // anywhere after a store would also wait for that store).
    v4 &w_sh_o = out.sh_o, &w_sh_d = out.sh_d, &w_mis_o = out.mis_o, &w_mis_d = out.mis_d, &w_nee0 = out.nee0, &w_nee1 = out.nee1, &w_ro = out.ro, &w_rd = out.rd, &w_beta = out.beta; u4 &w_nee2 = out.nee2; uint32_t w_stz = stv.z; float &w_cz = out.pre_z;
    bool w_skip = false; // null-BSDF skip: only the ray origin and the state word change
    w_sh_o = w_sh_d = w_mis_o = w_mis_d = w_nee0 = w_nee1 = w_ro = w_rd = w_beta = mkv4(splat3(0.0f), 0.0f); w_nee2.x = w_nee2.y = w_nee2.z = w_nee2.w = 0;
    BsdfT<MatLobes<MAT>::N> bsdf;
    if (!make_bsdf<MAT, FEAT>(sc, T.material, s, bsdf)) { // integrator.rs:434-439 (Q7)
        f3 o2 = spawn_origin(s.p, s.p_error, s.n, rd);
        w_ro = mkv4(o2, PT_INF);
        w_stz = st_pack(dim0, stv.z & ST_SPECULAR, bounces - 1);
        w_skip = true;
    } else {
    PT_STAMP(3, f2u(bsdf.ss.x) + f2u(bsdf.ts.y) + f2u(s.p.x) + f2u(s.p_error.x))
    const SpawnPair sp = spawn_pair(s.p, s.p_error, s.n); // every ray and pdf query of this vertex leaves from one of two offset points
    // ---- direct lighting: uniform_sample_one_light + estimate_direct up to the scene queries ----
    const uint32_t NS = BSDF_ALL & ~BSDF_SPECULAR;
#if defined(PTRS_ABL_SHADE_HALF) && PTRS_ABL_SHADE_HALF == 2
    const bool do_nee = false; // diagnostic build (tools/ablate.sh; timing only, wrong radiance): the vertex WITHOUT its next-event estimation
#else
    const bool do_nee = bsdf_num(bsdf, NS) > 0 && sc.n_lights > 0;
#endif
    if (do_nee != nee_guess) draw_vertex(X, S, stv, false, D); // only a lobeless Substrate gets here (do_nee false, nee_guess true): its draws start at the vertex's first dimension
    if (do_nee) {
        DLight Lt; // a register copy of the light's record (LDS-resident for scenes with few lights): the fields its kind reads are fetched together
        X.light(sc, li, Lt);
        const f2 u_light = mk2(u_nee[0], u_nee[1]), u_scat = mk2(u_nee[2], u_nee[3]);
        const bool delta = light_is_delta(Lt);
        LightSample ls;
        if ((FEAT & FEAT_INFINITE) && X.presampled(li)) { // inf_light_sample(u_light) of this vertex, evaluated by k_env_presample: the same function of the same numbers
            ls.p1_err = splat3(0.0f); ls.p1_n = splat3(0.0f);
            ls.wi = xyz(in.pre0); ls.pdf = in.pre0.w; ls.li = xyz(in.pre1);
            ls.p1 = f2u(in.pre1.w) ? s.p + ls.wi * (2.0f * Lt.world_radius) : s.p;
        } else light_sample_li<FEAT, CTX::inf_fallback>(sc, Lt, s.p, sp, u_light, ls, X.inf_marginal(li));
        PT_STAMP(4, f2u(ls.pdf) + f2u(ls.wi.x) + f2u(ls.li.x) + f2u(ls.p1.x))
        f3 A = splat3(0.0f);
        float spdf = 0.0f;
        f3 wi = ls.wi;
        if (ls.pdf > 0.0f && !is_black(ls.li)) {
            f3 f = bsdf_f(bsdf, wo, wi, NS) * fabs_(dot(wi, s.ns));
            spdf = bsdf_pdf(bsdf, wo, wi, NS);
            if (!is_black(f)) {
                // VisibilityTester::unoccluded -> spawn_ray_to_it (interaction.rs:50-60, Q13)
                f3 origin = spawn_from(sp, ls.p1 - s.p);
                f3 target = offset_ray_origin(ls.p1, ls.p1_err, ls.p1_n, origin - ls.p1);
                w_sh_o = mkv4(origin, PT_SHADOW_TMAX);
                w_sh_d = mkv4(target - origin, 0.0f);
                A = delta ? f * ls.li / ls.pdf : f * ls.li * power_heuristic(ls.pdf, spdf) / ls.pdf;
                out.shadow = true;
            }
        }
        PT_STAMP(5, f2u(A.x) + f2u(spdf) + (out.shadow ? 1u : 0u))
        f3 fB = splat3(0.0f); float wB = 1.0f;
        if (!delta) {
            uint32_t sampled = BSDF_ALL;
            fB = bsdf_sample_f(bsdf, wo, wi, u_scat, spdf, NS, sampled);
            fB = fB * fabs_(dot(wi, s.ns));
            if (!is_black(fB) && spdf > 0.0f) {
                bool ok = true;
                if (!(sampled & BSDF_SPECULAR)) {
                    float lpdf = light_pdf_li<FEAT>(sc, Lt, s.p, sp, wi);
                    if (lpdf == 0.0f) ok = false; // `return ld` (Q11)
                    else wB = power_heuristic(spdf, lpdf);
                }
                if (ok) {
                    w_mis_o = mkv4(spawn_from(sp, wi), PT_INF);
                    w_mis_d = mkv4(wi, 0.0f);
                    out.mis = true;
                }
            }
        }
        PT_STAMP(6, f2u(fB.x) + f2u(wB) + (out.mis ? 1u : 0u))
        if (out.shadow || out.mis) {
            out.nee = true;
            w_nee0 = mkv4(A, wB);
            w_nee1 = mkv4(fB, spdf);
            w_nee2.x = f2u(beta.x); w_nee2.y = f2u(beta.y); w_nee2.z = f2u(beta.z);
            w_nee2.w = li | ((out.shadow ? NEE_SHADOW : 0u) | (out.mis ? NEE_MIS : 0u)) << 24;
            if (!out.mis) { // shadow ray only (always, for delta lights): resolve_item's arithmetic with its one unknown, the occlusion, left open
                f3 ld = splat3(0.0f);
                ld = ld + A;
                // (two of its three floats ride in free slots: sh_d.w, sh_o.w -- a shadow ray's t_max is the constant 1 - 1e-4, nobody reads it
                // from there -- the third in pre_z; the record's kind travels in the queue entry (NEE_Q_PRE).  A shadow-only record is 36 bytes
                // in three stores instead of 80 in five.)
                const f3 c = beta * ((float)sc.n_lights * ld);
                w_sh_d.w = c.x; w_sh_o.w = c.y; w_cz = c.z;
            }
        }
    }

    // ---- continuation: integrator.rs:449-499 ---------------------------------------------------
    uint32_t dim = D.after_cont;
    f3 wi = splat3(0.0f);
    float pdf = 0.0f; uint32_t flags = 0;
#if defined(PTRS_ABL_SHADE_HALF) && PTRS_ABL_SHADE_HALF == 1
    f3 f = splat3(0.0f); // diagnostic build: the vertex WITHOUT its continuation (every path ends here)
#else
    f3 f = bsdf_sample_f(bsdf, wo, wi, mk2(u_tail[0], u_tail[1]), pdf, BSDF_ALL, flags);
#endif
    bool alive = true;
    if (D.err_cont) { out.err_dim = true; alive = false; } // a draw outside the sampler's dimensions: the reference panics (Sobol') / leaves the tables (stratified)
    if (is_black(f) || pdf == 0.0f) alive = false;
    if (alive) {
        beta = beta * (f * fabs_(dot(wi, s.ns)) / pdf);
        const bool spec = (flags & BSDF_SPECULAR) != 0;
        if (spec && (flags & BSDF_TRANSMISSION)) {
            float eta = bsdf.eta;
            eta_scale *= dot(wo, s.n) > 0.0f ? eta * eta : 1.0f / (eta * eta);
        }
        f3 o2 = spawn_from(sp, wi);
        if (R.rr_enable) {
            float mx = max_comp(beta * eta_scale);
            if (mx < R.rr_threshold && bounces > R.rr_start_depth) {
                float q = max_nz(0.05f, 1.0f - mx);
                dim = D.after_rr; // get_1d
                if (D.err_rr) { out.err_dim = true; alive = false; }
                else if (u_tail[2] < q) alive = false;
                else beta = beta / (1.0f - q);
            }
        }
        if (alive) {
            PT_STAMP(7, f2u(beta.x) + f2u(o2.x) + f2u(wi.x))
            bounces += 1;
            w_ro = mkv4(o2, PT_INF);
            w_stz = st_pack(dim, spec ? ST_SPECULAR : 0u, bounces);
            w_rd = mkv4(wi, u2f(w_stz));
            w_beta = mkv4(beta, eta_scale);
            out.next = true;
        }
    }
    if (out.err_dim) { out.nee = false; out.shadow = false; out.mis = false; out.next = false; }
    }
    if (w_skip) { w_rd = mkv4(rd, u2f(w_stz)); w_beta = bv; out.next = true; } // (the throughput moves with the ray: both are double-buffered by the round's parity)
    return out;
}

// The two scene queries of estimate_direct and its use of their results: shadow any-hit
// (integrator.rs:66-78, light.rs:38-42), MIS closest hit (119-134), then `l += beta * nLights * ld`
// (444-446, 206-216).  One NEE record per call.
// The part of estimate_direct after its two scene queries: `occluded` is the shadow ray's answer, `mh` the MIS ray's
// closest hit (prim < 0: it escaped).
template <int FEAT>
PT_HD void resolve_item(const DScene &sc, const DPaths &P, uint32_t entry, uint32_t f, bool occluded, const HitRec &mh) { // f: the entry's position in the NEE queue, where its record is
    const uint32_t pid = entry & NEE_Q_PID;
    if (entry & NEE_Q_PRE) { // the shade stage has done the arithmetic below for the unoccluded case (beta * nLights * ld in sh_d.w, sh_o.w, pre_z)
        if (!occluded) { const f3 c = mk3(pslot(P.sh_d, f).w, pslot(P.sh_o, f).w, pslot(P.pre_z, f)); const v4 Lv = pslot(P.L, pid); pslot(P.L, pid) = mkv4(xyz(Lv) + c, Lv.w); }
        return;
    }
    const u4 n2 = pslot(P.nee2, f);
    const uint32_t li = n2.w & 0xffffffu, fl = n2.w >> 24;
    const v4 n0 = pslot(P.nee0, f), n1 = pslot(P.nee1, f);
    f3 ld = splat3(0.0f);
    if ((fl & NEE_SHADOW) && !occluded) ld = ld + xyz(n0);
    if (fl & NEE_MIS) {
        const DLight &Lt = sc.lights[li];
        const f3 wi = xyz(pslot(P.mis_d, f));
        f3 l2 = splat3(0.0f);
        if (mh.prim >= 0) {
            const TriRegs T = load_tri_regs(sc.shade + mh.prim);
            if (T.light == (int32_t)li) { // std::ptr::eq(light, isect_light) (Q11)
                Surface s = tri_surface(T, mh.prim, mh.b0, mh.b1, mh.b2, -wi);
                l2 = surface_le<FEAT>(sc, T, s, -wi);
            }
        } else l2 = light_le<FEAT>(sc, Lt, wi);
        if (!is_black(l2)) ld = ld + xyz(n1) * l2 * splat3(1.0f) * n0.w / n1.w;
    }
    const f3 beta = mk3(u2f(n2.x), u2f(n2.y), u2f(n2.z));
    const v4 Lv = pslot(P.L, pid);
    const f3 L = xyz(Lv) + beta * ((float)sc.n_lights * ld);
    pslot(P.L, pid) = mkv4(L, Lv.w);
}

template <int FEAT, bool QUAD, class Stack, class Geom>
PT_HD void connect_item(const DScene &sc, const Geom &G, const DPaths &P, uint32_t entry, uint32_t f, Stack &stack, uint32_t &n_nodes, uint32_t &n_tris) {
    const uint32_t fl = ((entry & NEE_Q_SHADOW) ? (uint32_t)NEE_SHADOW : 0u) | ((entry & NEE_Q_MIS) ? (uint32_t)NEE_MIS : 0u) | ((entry & NEE_Q_PRE) ? (uint32_t)NEE_PRE : 0u);
    bool occluded = false;
    HitRec mh; mh.prim = -1; mh.t = 0.0f; mh.b0 = mh.b1 = mh.b2 = 0.0f; mh.flags = 0;
    if (fl & NEE_SHADOW) {
        const v4 o = pslot(P.sh_o, f), d = pslot(P.sh_d, f);
        HitRec h;
        occluded = bvh_trace_g<QUAD, true, (FEAT & FEAT_ALPHA) != 0>(G, sc, xyz(o), xyz(d), PT_SHADOW_TMAX, stack, h, n_nodes, n_tris); // (sh_o.w may hold a NEE_PRE record's payload)
    }
    if (fl & NEE_MIS) {
        const v4 o = pslot(P.mis_o, f);
        if (!bvh_trace_g<QUAD, false, (FEAT & FEAT_ALPHA) != 0>(G, sc, xyz(o), xyz(pslot(P.mis_d, f)), PT_INF, stack, mh, n_nodes, n_tris)) mh.prim = -1;
    }
    resolve_item<FEAT>(sc, P, entry, f, occluded, mh);
}

// run-time material dispatch (host twin; the HIP back end launches one specialised kernel per bucket)
template <int FEAT>
PT_HD ShadeResult shade_dispatch(int bucket, const DParams &R, const DSampler &S, const DCamera &C, const DScene &sc, const DPaths &P, uint32_t par, uint32_t pid, uint32_t e) {
    const PathIn in = load_path_in(P, par, pid, e);
    switch (bucket) {
        case 0: return shade_item<0, FEAT>(R, S, C, sc, P, pid, in, ShadeCtx());
        case 1: return shade_item<1, FEAT>(R, S, C, sc, P, pid, in, ShadeCtx());
        case 2: return shade_item<2, FEAT>(R, S, C, sc, P, pid, in, ShadeCtx());
        case 3: return shade_item<3, FEAT>(R, S, C, sc, P, pid, in, ShadeCtx());
        case 4: return shade_item<4, FEAT>(R, S, C, sc, P, pid, in, ShadeCtx());
        default: return shade_item<5, FEAT>(R, S, C, sc, P, pid, in, ShadeCtx());
    }
}

// One sample's contribution to output pixel (x, y): FilmTile::add_sample (film.rs:60-106) seen from the
// pixel.  pf = p_film of the sample, L its radiance; returns false when the pixel is outside the sample's
// filter footprint.
// (the weight of a pixel inside the sample's footprint; pdx = p_film.x - 0.5, pdy likewise)
PT_HD float film_weight_inside(float pdx, float pdy, int32_t x, int32_t y, const float *table) {
    const float radius = 2.0f, inv_r = 1.0f / radius;
    const float fx = fabs_(((float)x - pdx) * inv_r * 16.0f);
    const float fy = fabs_(((float)y - pdy) * inv_r * 16.0f);
    int32_t ix = (int32_t)floor_(fx); if (ix > 15) ix = 15;
    int32_t iy = (int32_t)floor_(fy); if (iy > 15) iy = 15;
    return table[iy * 16 + ix];
}
PT_HD bool film_weight(float pfx, float pfy, int32_t x, int32_t y, const float *table, float &w) {
    const float radius = 2.0f;
    const float pdx = pfx - 0.5f, pdy = pfy - 0.5f;
    const int32_t p0x = (int32_t)ceil_(pdx - radius), p0y = (int32_t)ceil_(pdy - radius);
    const int32_t p1x = (int32_t)(floor_(pdx + radius) + 1.0f), p1y = (int32_t)(floor_(pdy + radius) + 1.0f);
    if (x < p0x || x >= p1x || y < p0y || y >= p1y) return false;
    w = film_weight_inside(pdx, pdy, x, y, table);
    return true;
}

// Film gather for one output pixel over the current pass (host-twin form; the gfx950 kernel k_film does
// the same sums in the same order from LDS tiles).  table = 16x16 Gaussian filter table (film.rs:133-144).
// Order of accumulation: sample index outermost, then sample-pixel x, then y -- deterministic and
// independent of bands / tiles.  (The reference's own order depends on tile scheduling, film.rs:213-228.)
PT_HD void film_item(const DParams &R, const DSampler &S, const DPaths &P, const float *table, v4 *film, int32_t x, int32_t y) {
    v4 acc = film[(size_t)y * (size_t)R.W + (size_t)x];
    const uint32_t npix = (uint32_t)(R.row1 - R.row0) * (uint32_t)R.NX;
    const uint32_t ns = R.s1 - R.s0;
    for (uint32_t k = 0; k < ns; ++k) {
        for (int32_t qx = x - 2; qx <= x + 2; ++qx) {
            const int32_t sx = qx - S.min_x;
            if (sx < 0 || sx >= R.NX) continue;
            for (int32_t qy = y - 2; qy <= y + 2; ++qy) {
                const int32_t sy = qy - S.min_y;
                if (sy < R.row0 || sy >= R.row1) continue;
                const uint32_t pid = k * npix + (uint32_t)(sy - R.row0) * (uint32_t)R.NX + (uint32_t)sx;
                const f2a pf = pslot(P.pfilm, pid);
                float w;
                if (!film_weight(pf.x, pf.y, x, y, table, w)) continue;
                const v4 Lv = pslot(P.L, pid);
                acc.x += Lv.x * w; acc.y += Lv.y * w; acc.z += Lv.z * w; acc.w += w;
            }
        }
    }
    film[(size_t)y * (size_t)R.W + (size_t)x] = acc;
}

} // namespace pt
