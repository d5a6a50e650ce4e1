// pt_scene.h -- device-resident scene, sampler, camera and path-state layouts (all POD).
//
// HBM layout (DESIGN.md "Data layout"):
//   nodes   : 32-byte BVH nodes, depth-first (first child = i+1), two 16-byte loads per node.
//   tris    : 48-byte leaf-ordered triangle records (3 x 16-byte loads, no index indirection).
//   shade   : 160-byte per-triangle shading records (positions, normals, uvs, tangents, ids).
//   paths   : structure-of-arrays of 16-byte vectors in queue order (what travels with a path) or indexed by path slot (what it keeps); queues of slot ids.
#pragma once
#include "pt_vec.h"

namespace pt {

struct alignas(16) DNode { // = PtrsBvhNode = LinearBVHNode (accelerator.rs:89-95)
    float pmin[3];
    float pmax0; // pmax.x
    float pmax1, pmax2;
    uint32_t offset;    // leaf: first triangle record; interior: second child
    uint32_t meta;      // num_prims (low 16 bits) | axis << 16
};

// Traversal node: one per INTERIOR node of the binary tree, holding both children's boxes and
// references (4 x 16 B).  A reference is either an interior index or, with bit 31 set, a leaf:
// bits 0..26 first triangle record, bits 27..30 triangle count - 1.  One fetch tests two boxes and
// leaf nodes are never fetched.  axis = split axis of this node (near child = second iff dir[axis] < 0,
// exactly the reference's visiting order, accelerator.rs:397-408).
struct alignas(16) DNode2 {
    float c0min[3]; float c0max0;      // v0
    float c0max1, c0max2, c1min0, c1min1; // v1
    float c1min2; float c1max[3];      // v2
    uint32_t ref0, ref1, axis, pad;    // v3
};
static_assert(sizeof(DNode2) == 64, "node2");

// Quad node: TWO levels of the binary tree in one 128-byte record (8 x 16 B = one L2 line).  Slots 0,1 are the
// children of the node's first child A, slots 2,3 those of its second child B; when A (or B) is itself a leaf it
// occupies slot 0 (or 2) and slot 1 (or 3) is REF_NONE.  The boxes are stored PER AXIS: box[(2 a + u) * 4 + s] = lower (u = 0) / upper
// (u = 1) plane of slot s on axis a -- six vectors of four slots each, so that a ray reads, per axis, the vector of the planes it meets
// first and the vector of those it meets last (by the sign of its direction) instead of selecting 24 planes after the fetch.
// axes: bits 0-1 split axis of the node, 2-3 of A, 4-5 of B (3 = never swap); bit 8: entries stacked from this node
// are popped without the entry-distance re-test (chunks of one over-long leaf share the leaf's box); bits 12-14: number of
// occupied slots.  An empty slot (ref REF_NONE) holds the box (+inf, -inf), which no ray enters.
// Visiting [near group: near slot, far slot][far group: near slot, far slot] reproduces the binary traversal's
// order (accelerator.rs:397-408) with half the dependent fetches.
struct alignas(16) DNode4 {
    float box[24];
    uint32_t ref[4];
    uint32_t axes, pad[3];
};
static_assert(sizeof(DNode4) == 128, "node4");
#ifndef PTRS_QUAD_TOP_NODES
#define PTRS_QUAD_TOP_NODES 85
#endif
enum : uint32_t { QUAD_TOP_NODES = PTRS_QUAD_TOP_NODES }; // 1 + 4 + 16 + 64 records of the quad tree's top, kept in LDS by the traversal kernels (10.6 KB)
enum : uint32_t { PAIR_FORM_MAX_V4 = 1536 }; // largest scene the traversal kernels stage into LDS, in 16-byte vectors of its LDS form: 7 per pair node (its planes per axis and ray sign), 9 per triangle (three permuted copies)
enum : uint32_t { REF_LEAF = 0x80000000u, REF_NONE = 0xffffffffu, REF_FIRST_MASK = 0x07ffffffu, REF_COUNT_SHIFT = 27, REF_MAX_LEAF = 16 };

enum : uint32_t { TRI_HAS_NORMAL = 1, TRI_HAS_TANGENT = 2, TRI_REVERSE = 8, TRI_SWAPS = 16, TRI_DEGENERATE = 32, TRI_HAS_ALPHA = 64,
                  TRI_IS_LIGHT = 128,        // the triangle carries a DiffuseAreaLight
                  TRI_BUCKET_SHIFT = 8 };    // bits 8..10: material bucket (kind of the innermost material)

struct alignas(16) DTri { // leaf order; read by traversal as 3 x 16 B
    float p0[3]; float p1x;
    float p1y, p1z; float p2x, p2y;
    float p2z; uint32_t prim; uint32_t flags; int32_t alpha_tex;
};

// Per-triangle shading record, indexed by global triangle id (mesh-major).  Thirteen 16-byte vectors so
// the shade stage fetches it with wide loads (v0-v6 always, v7-v9 only when the mesh has tangents, v10-v11 always,
// v12 only where texture differentials are computed).  v10-v12 hold what Triangle::intersect (shape.rs:187-266)
// recomputes at every hit although it depends on the triangle alone -- the geometric normal, dp/du, dp/dv and
// normalize(dp/du) -- evaluated once on the host with the same binary32 operations (pt_tri.h).
struct alignas(16) DTriShade {
    float p0[3]; float p1x;                                   // v0
    float p1y, p1z, p2x, p2y;                                 // v1
    float p2z; int32_t material; int32_t light; uint32_t flags; // v2  light = index into lights[] or -1
    float n0[3]; float n1x;                                   // v3
    float n1y, n1z, n2x, n2y;                                 // v4
    float n2z; int32_t alpha_tex; float uv0[2];               // v5
    float uv1[2], uv2[2];                                     // v6
    float s0[3]; float s1x;                                   // v7
    float s1y, s1z, s2x, s2y;                                 // v8
    float s2z; uint32_t pad[3];                               // v9
    float ng[3]; float ssn0;                                  // v10 ng = normalize(cross(p0-p2, p1-p2)), flipped for reverse_orientation ^ swaps_handedness
    float ssn1, ssn2, dpdu0, dpdu1;                           // v11 ssn = normalize(dpdu)
    float dpdu2; float dpdv[3];                               // v12
};
static_assert(sizeof(DNode) == 32, "node");
static_assert(sizeof(DTri) == 48, "tri");
static_assert(sizeof(DTriShade) == 208, "trishade");

// tex[k] >= 0: texture id.  When that texture is a ConstantTexture its value is also folded into
// cval[k] and bit k of const_mask is set, which saves the dependent texture-record fetch.
struct alignas(16) DMaterial {
    int32_t kind; int32_t flags; int32_t inner; uint32_t const_mask; // v0
    int32_t tex[6]; int32_t pad[2];                                  // v1, v2(half)
    float cval[6][4];                                                // rgb (or value in .x) per slot
};
struct DTexLevel { uint64_t offset; int32_t cols, rows; };
struct DTexture {
    int32_t kind, channels;
    float value[3], value2[3];
    float su, sv, du, dv;
    int32_t wrap, n_levels;
    uint32_t first_level; uint32_t pad;
};
struct alignas(16) DLight {
    int32_t kind; int32_t tri; int32_t ke_tex; float area;      // v0
    float v[3]; float world_radius;                             // v1
    float c[3]; uint32_t ke_const;                              // v2  c = I / L, or the emission when ke is constant (ke_const = 1)
    DTriShade T;                                                // area lights: a copy of the light's triangle record
    int32_t lmap_tex; int32_t nu, nv; float marg_int;
    uint32_t func_off, cdf_off, fint_off, mcdf_off;
    // guide tables for the two cdf searches (0 = none): entry k of a table with g cells = the number of leading cdf
    // values <= k/g, so the search for u only has to look between entries floor(u*g) and floor(u*g)+1
    uint32_t guide_u, guide_v, cguide_off, mguide_off;
    // area lights: the normals Triangle::sample and pdf_at_point's intersection end up with, when they do not depend on
    // where the triangle is hit (n_ok; checked at build time): both are +-(a geometric normal), the sign decided by
    // face-forwarding to the interpolated shading normal
    float n_sample[3]; uint32_t n_ok; float n_point[3]; uint32_t pad_n;
    float l2w[12], w2l[12];
};

struct DScene {
    const DNode2 *nodes2;   // traversal nodes, pair form: small scenes that the kernels stage into LDS
    uint32_t n_nodes2, n_nodes4; // exactly one of the two forms is present
    const DNode4 *nodes4;   // traversal nodes, quad form: everything else
    const DNode *nodes;     // the binary tree in the reference's 32-byte layout (kept for stats / export)
    const DTri *tris;
    const DTriShade *shade;
    const DMaterial *mats;
    const DTexture *texs;
    const DTexLevel *levels;
    const float *texdata;
    const DLight *lights;
    const float *distdata;
    const uint32_t *inf_lights;
    uint32_t n_nodes, n_prims, n_lights, n_inf;
};

enum { SOBOL_TAB_DIMS = 256, SOBOL_NIBBLES = 13 }; // 13 nibbles hold the 52 matrix columns

struct DSampler { // SobolSamplerBuilder::new (sobol.rs:35-60) + table rows
    const uint32_t *matrices; // [1024*52]
    const uint32_t *bytetab;  // optional [SOBOL_TAB_DIMS][8][256]: XOR of the matrix columns selected by one index byte
    const uint32_t *nibtab;   // optional [1024][SOBOL_NIBBLES][16]: the same per index nibble (what the shade kernels stage into LDS)
    const uint64_t *vdc;      // row m-1 of VD_C_SOBOL_MATRICES
    const uint64_t *vdc_inv;  // row m-1 of VD_C_SOBOL_MATRICES_INV
    uint32_t log2_res;
    int32_t resolution;
    int32_t min_x, min_y; // sample bounds p_min
    uint32_t spp;         // Sobol': power of two; stratified: dim_pixel_samples^2
    // PTRS_SAMPLER_STRATIFIED (pt_stratified.h): per-pixel tables made by k_strat_tables before the passes
    uint32_t kind, strat_dims;  // kind = PTRS_SAMPLER_*; n_sampled_dimensions
    const float *strat1;        // [NX*NY][strat_dims][spp]
    const float *strat2;        // [NX*NY][strat_dims][spp][2]
};

struct DCamera { // = PtrsCamera
    float rot[4], trans[3];
    float m00, m11, m22, m23;
    float r2s[16];
    float dxc[3], dyc[3];
};

struct DParams {
    int32_t max_depth;
    float rr_threshold;
    int32_t rr_start_depth, rr_enable;
    int32_t NX, NY;        // sample-bounds extent (W+4, H+4)
    int32_t W, H;          // film resolution
    float inv_sqrt_spp;    // 1 / sqrt(spp as f32)
    // current pass: sample rows [row0,row1) of the sample grid, samples [s0,s1)
    int32_t row0, row1;
    uint32_t s0, s1;
    uint32_t n_paths;      // (row1-row0)*NX*(s1-s0)
    uint32_t counters_on;
    int32_t pixel_mode, pix_sx, pix_sy; // render_single_pixel: the pass is the samples [s0,s1) of sample-pixel (pix_sx, pix_sy) and nothing else
    uint32_t *row_cost;    // ptrs_render_row_cost: one counter per sample row of the grid (NY), + 1 for every BVH query a path of that row makes; null in a render
};

struct alignas(16) v4 { float x, y, z, w; };
struct alignas(16) u4 { uint32_t x, y, z, w; };
struct alignas(8) f2a { float x, y; }; // an 8-byte element of a device array (one load / store)

// path-state bits (u4.z of `st`)
enum : uint32_t { ST_DIM_MASK = 0xfffu, ST_SPECULAR = 1u << 12, ST_HAS_DIFF = 1u << 13, ST_BOUNCE_SHIFT = 16 };
// An entry of the NEE queue: the path slot (bits 0-26: a pass holds at most 2^27 paths) and what the record holds -- a shadow ray (bit 29),
// a MIS ray (bit 30), shadow-only with the contribution precomputed (bit 31, NEE_PRE).  The connect stage knows which rays to fetch
// from the entry alone (no flags word to load first), and for a shadow-only record it reads nothing of nee0 / nee1 / nee2.
enum : uint32_t { NEE_Q_PID = 0x07ffffffu, NEE_Q_SHADOW = 0x20000000u, NEE_Q_MIS = 0x40000000u, NEE_Q_PRE = 0x80000000u };
// nee flags (stored in nee2.w as bits)
enum : uint32_t { NEE_SHADOW = 1u, NEE_MIS = 2u,
                  NEE_PRE = 4u,          // shadow-only record whose contribution beta * nLights * ld is already in (sh_d.w, sh_o.w, pre_z): the connect stage adds it when the ray is free
                  NEE_OCCLUDED = 0x80u }; // set by the split connect stage when the shadow ray was blocked

// Path state.  What a path carries from vertex to vertex travels IN QUEUE ORDER: the ray, its throughput and the hit sit at the
// path's position `e` in the round's extension queue (e = segment * seg_cap + index: the index of the queue entry itself), the records of
// a pending next-event estimation at the position `f` of its entry in the NEE queue.  The wave that owns a segment writes the
// positions it hands out with wave_push -- consecutive ones: 64 lanes x 16 bytes are eight whole 128-byte lines -- and the stage
// that consumes the queue reads them in order; round 2 indexed all of it by path slot, where the lanes of a wave touch as many
// lines as paths (the slots of a segment's surviving paths drift apart as paths end).  What a path keeps for good stays at its slot
// `pid`: the Sobol' index and scramble, the film position, the radiance.  The arrays written by one round's shade stage and read by
// the next round's (ray, throughput) are double-buffered by the round's parity, like the extension queue.
// (integrator.rs:399-405: l, beta, ray, bounces, specular_bounce, eta_scale)
struct DPaths {
    v4 *ray_o[2]; // [parity][e]: o.xyz, (t_max of an extension ray: +inf; nobody reads it from here)
    v4 *ray_d[2]; // [parity][e]: d.xyz, the path's state word (dimension counter | flags | bounces): rewritten with the direction by every vertex that continues
    v4 *beta[2];  // [parity][e]: beta.rgb, eta_scale
    u4 *hit;      // [e]: prim (int), b0, b1, b2 (float bits) of the round's extension ray
    v4 *pre0;     // [e]: environment-lit scenes, the vertex's presampled light sample (k_env_presample): wi.xyz, pdf
    v4 *pre1;     // [e]: Li.rgb, valid
    v4 *L;        // [pid]: L.rgb, 0
    u4 *st;       // [pid]: what never changes along a path: sobol index lo, hi (stratified: pixel, sample), (unused), pixel scramble -- written once by k_generate
    f2a *pfilm;   // [pid]: p_film.xy (8 bytes per path) -- written by generate, read by the film kernel
    v4 *nee0;     // [f]: records with a MIS ray: A.rgb (light-sampling term, final if unoccluded), weight of the BSDF term
    v4 *nee1;     // [f]: f.rgb of the BSDF term (already times |wi.ns|), scattering pdf
    u4 *nee2;     // [f]: records with a MIS ray: beta at the vertex (rgb bits), light index | flags << 24
    v4 *sh_o;     // [f]: shadow ray o.xyz (its t_max is the constant PT_SHADOW_TMAX); NEE_PRE records: contribution .y in w
    v4 *sh_d;     // [f]: shadow ray d.xyz; NEE_PRE records: contribution .x in w
    float *pre_z; // [f]: NEE_PRE records: contribution .z
    v4 *mis_o;    // [f]: MIS ray o.xyz
    v4 *mis_d;    // [f]: MIS ray d.xyz
    u4 *nhit;     // [f]: the MIS ray's closest hit (connect stage -> resolve)
};

// queue counters: one row of uint32 per loop iteration
enum { Q_EXT = 0, Q_SHADOW = 1, Q_MIS = 2, Q_NEE = 3, Q_MAT0 = 4, Q_NUM_MAT = 7, Q_STRIDE = 16 };
enum { CNT_EXT = 0, CNT_SHADOW = 1, CNT_MIS = 2, CNT_NODES = 3, CNT_TRIS = 4,
       CNT_NODE_STEPS = 5, CNT_NODE_VISITS = 6, CNT_TRI_STEPS = 7, // refill kernels with PTRS_FLAG_COUNTERS: wave-level node / triangle steps x 64 and per-lane node visits (lane occupancy of each phase)
       CNT_ERR = 8,                                                // PTRS_ERRFLAG_* raised by device code
       CNT_STAMP0 = 16,                                            // diagnostic builds (-DPTRS_STAMPS): 12 phase clocks of k_shade
       CNT_NUM = 32 };

struct alignas(8) MatEntry { uint32_t e, pid; }; // a shade-queue entry: the vertex's position in the round's extension queue (where its ray and hit are) and its path slot
struct DQueues {
    uint32_t *ext[2];          // ping-pong extension-ray queues
    MatEntry *mat[Q_NUM_MAT];  // one shade queue per material kind
    uint32_t *nee;             // paths with a pending next-event-estimation record this round
    uint32_t *counts;          // [iters][Q_STRIDE][G]
    uint32_t *tickets;         // [iters][Q_STRIDE][...]: segment tickets of the persistent queue kernels, one set of counters per launch of a pass
    uint32_t *alive;           // [iters]: round i still has paths (set by the shade kernels of round i - 1)
    unsigned long long *stats; // [CNT_NUM]
};

// Element `i` of a device array through a 32-BIT byte offset (i * sizeof(T) < 2^32: a pass holds at most 2^27 paths of <= 16 bytes, a queue
// at most 2^27 + G * 64 entries of 4): on gfx950 the access is `global_load ... v_offset, s[base]` with one shift shared by all arrays of the
// element size, instead of a 64-bit shift-and-add per access (v_lshl_add_u64: an instruction of the half-rate class, 8 % of the Matte shade
// kernel's) and an address register pair each.
template <class T> PT_HD T &pslot(T *base, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + (size_t)(i * (uint32_t)sizeof(T)));
#else
    return base[i];
#endif
}
template <class T> PT_HD const T &pslot(const T *base, uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(i * (uint32_t)sizeof(T)));
#else
    return base[i];
#endif
}

// A store of path state (on its own so that store policies can be tried in one place: streaming / non-temporal stores measured
// +2 % on the Cornell frame, -2 % on colonnade, +0.7 % on classroom, and were left out)
template <class T> PT_HD void pstore(T *base, uint32_t i, const T &v) { pslot(base, i) = v; }

PT_HD uint32_t f2u(float f) { return ptf_bits(f); }
PT_HD float u2f(uint32_t u) { return ptf_from_bits(u); }
PT_HD v4 mkv4(f3 a, float w) { v4 r; r.x = a.x; r.y = a.y; r.z = a.z; r.w = w; return r; }
PT_HD f3 xyz(v4 a) { return mk3(a.x, a.y, a.z); }
PT_HD f3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }

// register copy of a DTriShade, filled by wide loads
struct TriRegs {
    f3 p0, p1, p2, n0, n1, n2, s0, s1, s2;
    f3 ng, ssn, dpdu, dpdv; // per-triangle constants (DTriShade v10-v12)
    f2 uv0, uv1, uv2;
    int32_t material, light, alpha_tex; uint32_t flags;
};
PT_HD TriRegs load_tri_regs(const DTriShade *rec, bool want_dp = true) {
    const v4 *q = reinterpret_cast<const v4 *>(rec);
    TriRegs t;
    const v4 k = q[10], l = q[11];
    t.ng = mk3(k.x, k.y, k.z); t.ssn = mk3(k.w, l.x, l.y);
    if (want_dp) { const v4 m = q[12]; t.dpdu = mk3(l.z, l.w, m.x); t.dpdv = mk3(m.y, m.z, m.w); }
    else { t.dpdu = t.dpdv = mk3(0.0f, 0.0f, 0.0f); }
    const v4 a = q[0], b = q[1], c = q[2];
    t.p0 = mk3(a.x, a.y, a.z); t.p1 = mk3(a.w, b.x, b.y); t.p2 = mk3(b.z, b.w, c.x);
    t.material = (int32_t)f2u(c.y); t.light = (int32_t)f2u(c.z); t.flags = f2u(c.w);
    const v4 d = q[3], e = q[4], f = q[5], g = q[6];
    t.n0 = mk3(d.x, d.y, d.z); t.n1 = mk3(d.w, e.x, e.y); t.n2 = mk3(e.z, e.w, f.x);
    t.alpha_tex = (int32_t)f2u(f.y); t.uv0 = mk2(f.z, f.w); t.uv1 = mk2(g.x, g.y); t.uv2 = mk2(g.z, g.w);
    if (t.flags & TRI_HAS_TANGENT) {
        const v4 h = q[7], i = q[8], j = q[9];
        t.s0 = mk3(h.x, h.y, h.z); t.s1 = mk3(h.w, i.x, i.y); t.s2 = mk3(i.z, i.w, j.x);
    } else { t.s0 = t.s1 = t.s2 = mk3(0.0f, 0.0f, 0.0f); }
    return t;
}

} // namespace pt
