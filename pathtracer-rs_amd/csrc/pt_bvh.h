// pt_bvh.h -- BVH traversal over pair nodes (LDS-resident scenes) and quad nodes (everything else), both derived
// from the flattened 32-byte-node binary tree.
//
// Restates BVH::intersect / intersect_p (src/pathtracer/accelerator.rs:359-475) and
// Bounds3::intersect_p_precomp (src/common/bounds.rs:190-232): near-child-first by split axis,
// far child pushed, leaf triangles tested in order (a later hit with equal t replaces the
// earlier one, Q14).  The traversal stack is supplied by the caller: on gfx950 it lives in LDS
// (one column per lane, see kernels.hip), on the host twin it is a local array.
#pragma once
#include "pt_texture.h"

namespace pt {

struct HitRec { int32_t prim; float t, b0, b1, b2; uint32_t flags; };

// Where traversal reads geometry from.  GeomGlobal: the HBM arrays (through L1/L2).  GeomLocal:
// a 16-byte-vector copy [4 per pair node | 3 per triangle] -- on gfx950 the kernels stage small scenes
// (Cornell: 29 pair nodes + 36 triangles = 3.6 KB) into LDS once per workgroup and walk them there.
struct GeomGlobal {
    const v4 *nodesv; const DTri *tris; // nodesv: the scene's pair nodes (4 vectors each) or quad nodes (8 vectors each)
    PT_MEM void node(uint32_t i, v4 &a, v4 &b, v4 &c, v4 &d) const { const v4 *q = nodesv + 4u * i; a = q[0]; b = q[1]; c = q[2]; d = q[3]; }
    PT_MEM void node8(uint32_t i, v4 *o) const { const v4 *q = nodesv + 8u * i; for (int k = 0; k < 8; ++k) o[k] = q[k]; }
    // quad node i: the near and far plane vectors of a ray whose per-axis near-plane byte offsets are (px, py, pz) (0 | 16, 32 | 48, 64 | 80:
    // the far vector is the other one of the pair), the references, the axes word
    PT_MEM void quad_load(uint32_t i, uint32_t px, uint32_t py, uint32_t pz, v4 &xn, v4 &xf, v4 &yn, v4 &yf, v4 &zn, v4 &zf, v4 &refs, v4 &meta) const {
        const char *b = reinterpret_cast<const char *>(nodesv);
        const uint32_t off = i * 128u; // 32-bit byte offsets from one base: the six plane reads cost an add (and a xor) each, not a 64-bit address
        xn = *reinterpret_cast<const v4 *>(b + (size_t)(off + px)); xf = *reinterpret_cast<const v4 *>(b + (size_t)((off + px) ^ 16u));
        yn = *reinterpret_cast<const v4 *>(b + (size_t)(off + py)); yf = *reinterpret_cast<const v4 *>(b + (size_t)((off + py) ^ 16u));
        zn = *reinterpret_cast<const v4 *>(b + (size_t)(off + pz)); zf = *reinterpret_cast<const v4 *>(b + (size_t)((off + pz) ^ 16u));
        refs = *reinterpret_cast<const v4 *>(b + (size_t)(off + 96u)); meta = *reinterpret_cast<const v4 *>(b + (size_t)(off + 112u));
    }
    PT_MEM void tri(uint32_t k, v4 &a, v4 &b, v4 &c) const { const v4 *q = reinterpret_cast<const v4 *>(tris + k); a = q[0]; b = q[1]; c = q[2]; }
};
struct GeomLocal {
    const v4 *nodes4; const v4 *tris4;
    PT_MEM void node(uint32_t i, v4 &a, v4 &b, v4 &c, v4 &d) const { const v4 *q = nodes4 + 4u * i; a = q[0]; b = q[1]; c = q[2]; d = q[3]; }
    PT_MEM void node8(uint32_t i, v4 *o) const { const v4 *q = nodes4 + 8u * i; for (int k = 0; k < 8; ++k) o[k] = q[k]; }
    PT_MEM void quad_load(uint32_t i, uint32_t px, uint32_t py, uint32_t pz, v4 &xn, v4 &xf, v4 &yn, v4 &yf, v4 &zn, v4 &zf, v4 &refs, v4 &meta) const {
        const v4 *q = nodes4 + 8u * i;
        xn = q[px >> 4]; xf = q[(px >> 4) ^ 1u]; yn = q[py >> 4]; yf = q[(py >> 4) ^ 1u]; zn = q[pz >> 4]; zf = q[(pz >> 4) ^ 1u]; refs = q[6]; meta = q[7];
    }
    PT_MEM void tri(uint32_t k, v4 &a, v4 &b, v4 &c) const { a = tris4[3u * k]; b = tris4[3u * k + 1u]; c = tris4[3u * k + 2u]; }
};

// The decision of Bounds3::intersect_p_precomp (bounds.rs:190-232) from its six plane distances (the upper ends already scaled by
// 1 + 2 gamma(3)), gfx950 form.  The reference runs `if t_min > ty_max || ty_min > t_max { return false }`, takes the larger lower end and
// the smaller upper end, does the same with z, and ends with `t_min < ray.t_max && t_max > 0`.  Its four interval tests are the six
// comparisons "lower end of one axis > upper end of another", so
//     no test fails and t_max > 0   <=>   max3(lower ends) <= min3(upper ends) and min3(upper ends) > 0:
// the three same-axis pairs the right side adds cannot fail once every upper end is positive (lower <= upper on an axis, rounding is
// monotone, and scaling a positive upper end by k > 1 only raises it).  Two v_max3 / v_min3 and two comparisons instead of four
// v_max / v_min and five comparisons -- on gfx950 comparisons, selects and min / max all issue at half the rate of fp32 add / mul
// (tools/valu_ceiling.hip), they are what the slab test's time is made of.
// NaNs (0 * inf: the ray lies in one of the box's planes): the comparison form ignores a NaN that comes in from y or z (every comparison
// with it is false, the running values keep their old contents) and so do v_max3 / v_min3 (IEEE maxNum / minNum); a NaN on the x axis
// stays in the reference's running value to the end and fails the box (t_max > 0 is false for a NaN t_max; for a NaN t_min the entry
// distance is NaN and the caller's t_entry < ray.t_max is false): the explicit `ordered` test fails the box in exactly those cases
// (test_rays_inside_box_planes).  (+0 / -0 can come out differently in the entry distance; it is only ever compared.)
PT_HD bool slab_finish(float tx_min, float ty_min, float tz_min, float tx_max, float ty_max, float tz_max, float &t_entry) {
    const bool x_ordered = !__builtin_isunordered(tx_min, tx_max);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(tx_min, ty_min), tz_min);
    const float tf = __builtin_fminf(__builtin_fminf(tx_max, ty_max), tz_max);
    t_entry = tn;
    return (tn <= tf) & (tf > 0.0f) & x_ordered;
}
// Bounds3::intersect_p_precomp (bounds.rs:190-232) on explicit box corners, split in two: everything
// that does not depend on ray.t_max (the per-axis interval tests and `t_max_box > 0`) is decided here
// and the entry distance returned; the caller finishes the test with `t_entry < ray.t_max`.
// (slab_planes: the same from the planes the ray meets first / last on each axis -- bounds[dir_is_neg] / bounds[1 - dir_is_neg] of
// bounds.rs:196-203 -- already picked: the quad nodes keep their planes per axis, and a lane READS the near and the far vector of its
// ray's signs instead of selecting every plane after the fetch)
PT_HD bool slab_planes(float nx, float ny, float nz, float fx, float fy, float fz, f3 o, f3 inv, float &t_entry);
PT_HD bool slab_entry6(float minx, float miny, float minz, float maxx, float maxy, float maxz, f3 o, f3 inv, const bool neg[3], float &t_entry) {
    return slab_planes(neg[0] ? maxx : minx, neg[1] ? maxy : miny, neg[2] ? maxz : minz, neg[0] ? minx : maxx, neg[1] ? miny : maxy, neg[2] ? minz : maxz, o, inv, t_entry);
}
PT_HD bool slab_planes(float nx, float ny, float nz, float fx, float fy, float fz, f3 o, f3 inv, float &t_entry) {
    const float k = 1.0f + 2.0f * gamma_err(3);
    float t_min = (nx - o.x) * inv.x;
    float t_mx = (fx - o.x) * inv.x;
    const float ty_min = (ny - o.y) * inv.y;
    float ty_max = (fy - o.y) * inv.y;
    t_mx *= k; ty_max *= k;
    const float tz_min = (nz - o.z) * inv.z;
    float tz_max = (fz - o.z) * inv.z;
    tz_max *= k;
#if defined(__HIP_DEVICE_COMPILE__)
    return slab_finish(t_min, ty_min, tz_min, t_mx, ty_max, tz_max, t_entry);
#else
    // straight-line form of the early-out code (same comparisons in the same order, so the same answer also when a
    // product is NaN): 64 lanes rarely agree on which interval test fails, and a branch per test costs more than it saves
    const bool miss_y = (t_min > ty_max) | (ty_min > t_mx);
    t_min = ty_min > t_min ? ty_min : t_min;
    t_mx = ty_max < t_mx ? ty_max : t_mx;
    const bool miss_z = (t_min > tz_max) | (tz_min > t_mx);
    t_min = tz_min > t_min ? tz_min : t_min;
    t_mx = tz_max < t_mx ? tz_max : t_mx;
    t_entry = t_min;
    return !(miss_y | miss_z) & (t_mx > 0.0f);
#endif
}

// Alpha-mask test of an accepted candidate (shape.rs:227-244 / 470-521): the mask texture is looked up at
// the interpolated uv with zero differentials; a value of exactly 0 rejects the hit.
PT_HD bool alpha_rejects(const DScene &sc, uint32_t prim, int32_t alpha_tex, const TriHit &h) {
    const v4 *q = reinterpret_cast<const v4 *>(sc.shade + prim);
    const v4 f = q[5], g = q[6]; // uv0 in f.zw, uv1 in g.xy, uv2 in g.zw
    const f2 uv = mk2(h.b0 * f.z + h.b1 * g.x + h.b2 * g.z, h.b0 * f.w + h.b1 * g.y + h.b2 * g.w);
    return tex_eval<FEAT_FULL>(sc, alpha_tex, uv, 0.0f, 0.0f, 0.0f, 0.0f).x == 0.0f;
}

// ANY = false: closest hit (BVH::intersect); ANY = true: any hit (intersect_p), accelerator.rs:359-475.
// Stack must provide push(ref, t_entry), pop(ref&, t_entry&), empty() and clear().
//
// Per ray the leaves are visited in the reference's order: at an interior node the child on the
// near side of the split axis first, the other one postponed on the stack.  Differences of form:
//   * "while-while": a lane first descends to a leaf, then tests that leaf's triangles, so on a
//     64-lane wave the expensive triangle phase runs with most lanes active;
//   * both children's boxes are read with the parent (pair nodes).  The reference tests a postponed
//     child's box only when it is popped, against the t_max of that moment (accelerator.rs:372,
//     bounds.rs:231).  Of that test only the last comparison `t_entry < ray.t_max` depends on t_max, so
//     the postponed child is stacked together with its entry distance and the comparison is redone
//     when it is popped: the same subtrees are entered as in the reference, which matters when two
//     coincident surfaces compete at (rounded-)equal t.
// n_nodes counts child boxes tested, n_tris triangle tests.
template <bool ANY, class Stack>
PT_HD uint32_t pop_next_ref(Stack &stack, float t_max) {
    while (!stack.empty()) { uint32_t r; float te; stack.pop(r, te); if (ANY || te < t_max) return r; }
    return REF_NONE;
}

// One pair-node visit: tests both children, stacks the far one when both are hit and returns the next reference in `cur`.
template <bool ANY, class Stack, class Geom>
PT_HD void pair_visit(const Geom &G, uint32_t &cur, f3 o, f3 inv, const bool neg[3], float t_max, Stack &stack, uint32_t &n_nodes) {
    v4 a, b, c, e;
    G.node(cur, a, b, c, e);
    const uint32_t ref0 = f2u(e.x), ref1 = f2u(e.y), axis = f2u(e.z);
    n_nodes += 2;
    float t0 = 0.0f, t1 = 0.0f;
    const bool h0 = slab_entry6(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, neg, t0) & (t0 < t_max);
    const bool h1 = (ref1 != REF_NONE) & slab_entry6(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, neg, t1) & (t1 < t_max);
    const bool second_first = axis < 3u && neg[axis];
    // both hit: the near one is entered, the far one postponed; one hit: that one (whichever side it is on); none: the stack.
    // (Written on h0 / h1 themselves: a select between two lane masks would be computed on 0/1 integers in vector registers.)
    if (h0 & h1) { stack.push(second_first ? ref0 : ref1, second_first ? t0 : t1); cur = second_first ? ref1 : ref0; }
    else if (h0 | h1) cur = h0 ? ref0 : ref1;
    else cur = pop_next_ref<ANY>(stack, t_max);
}

// The leaf's triangles, in order (a later hit with equal t replaces the earlier one, Q14).  Returns true when an
// any-hit query (compile-time ANY, or any_rt for kernels that mix both kinds of rays in one loop) is done.
template <bool ANY, bool ALPHA, class Geom>
PT_HD bool leaf_test(const Geom &G, const DScene &sc, uint32_t leaf, f3 o, const RayShear &shear, float &t_max, HitRec &out, bool &hit, uint32_t &n_tris, bool any_rt = false) {
    const uint32_t leaf_first = leaf & REF_FIRST_MASK, leaf_count = ((leaf >> REF_COUNT_SHIFT) & 15u) + 1u;
    for (uint32_t i = 0; i < leaf_count; ++i) {
        v4 ta, tb, tc;
        G.tri(leaf_first + i, ta, tb, tc);
        const f3 p0 = mk3(ta.x, ta.y, ta.z), p1 = mk3(ta.w, tb.x, tb.y), p2 = mk3(tb.z, tb.w, tc.x);
        const uint32_t prim = f2u(tc.y), flags = f2u(tc.z);
        ++n_tris;
        TriHit h;
        if (tri_test_s(o, shear, t_max, p0, p1, p2, h) && !(flags & TRI_DEGENERATE)) {
            if (ALPHA && (flags & TRI_HAS_ALPHA) && alpha_rejects(sc, prim, (int32_t)f2u(tc.w), h)) continue;
            if (ANY || any_rt) { out.prim = 0; hit = true; return true; }
            hit = true; t_max = h.t;
            out.prim = (int32_t)prim; out.t = h.t; out.b0 = h.b0; out.b1 = h.b1; out.b2 = h.b2; out.flags = flags;
        }
    }
    return false;
}

// ONE triangle of a leaf (the phase-voting kernels take a leaf a triangle at a time): tests the first triangle of `leaf`
// and rewrites `leaf` to the rest of it -- REF_NONE when that was the last one.  Same tests in the same order as leaf_test.
// Returns true when an any-hit query is done.
template <bool ALPHA, class Geom>
PT_HD bool leaf_step(const Geom &G, const DScene &sc, uint32_t &leaf, f3 o, const RayShear &shear, float &t_max, HitRec &out, bool &hit, uint32_t &n_tris, bool any_rt) {
    const uint32_t first = leaf & REF_FIRST_MASK, rest = (leaf >> REF_COUNT_SHIFT) & 15u; // triangles after this one
    leaf = rest ? (REF_LEAF | ((rest - 1u) << REF_COUNT_SHIFT) | (first + 1u)) : REF_NONE;
    v4 ta, tb, tc;
    G.tri(first, ta, tb, tc);
    const f3 p0 = mk3(ta.x, ta.y, ta.z), p1 = mk3(ta.w, tb.x, tb.y), p2 = mk3(tb.z, tb.w, tc.x);
    const uint32_t prim = f2u(tc.y), flags = f2u(tc.z);
    ++n_tris;
    TriHit h;
    if (!ALPHA) { // select form (pt_tri.h: tri_test_perm_sel): no divergent region around the test's rejections and the hit record
        const bool ok = tri_test_s_sel(o, shear, t_max, p0, p1, p2, h) && !(flags & TRI_DEGENERATE);
        hit = hit | ok;
        t_max = ok ? h.t : t_max;
        out.prim = ok ? (any_rt ? 0 : (int32_t)prim) : out.prim; out.t = ok ? h.t : out.t; out.b0 = ok ? h.b0 : out.b0; out.b1 = ok ? h.b1 : out.b1; out.b2 = ok ? h.b2 : out.b2; out.flags = ok ? flags : out.flags;
        return ok & any_rt;
    }
    if (tri_test_s(o, shear, t_max, p0, p1, p2, h) && !(flags & TRI_DEGENERATE)) {
        if (ALPHA && (flags & TRI_HAS_ALPHA) && alpha_rejects(sc, prim, (int32_t)f2u(tc.w), h)) return false;
        if (any_rt) { out.prim = 0; hit = true; return true; }
        hit = true; t_max = h.t;
        out.prim = (int32_t)prim; out.t = h.t; out.b0 = h.b0; out.b1 = h.b1; out.b2 = h.b2; out.flags = flags;
    }
    return false;
}

template <bool ANY, bool ALPHA, class Stack, class Geom>
PT_HD bool bvh_trace_pair(const Geom &G, const DScene &sc, f3 o, f3 d, float t_max, Stack &stack, HitRec &out, uint32_t &n_nodes, uint32_t &n_tris) {
    out.prim = -1; out.t = t_max; out.b0 = out.b1 = out.b2 = 0.0f; out.flags = 0;
    stack.clear(); // an any-hit query may have returned early and left entries behind
    if (sc.n_nodes2 + sc.n_nodes4 == 0) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    const RayShear shear = ray_shear_inv(d, inv);
    uint32_t cur = 0; // reference to process next (interior index or leaf), REF_NONE when done
    bool hit = false;
    while (cur != REF_NONE) {
        while (cur != REF_NONE && !(cur & REF_LEAF)) pair_visit<ANY>(G, cur, o, inv, neg, t_max, stack, n_nodes); // descend to a leaf
        if (cur == REF_NONE) break;
        if (leaf_test<ANY, ALPHA>(G, sc, cur, o, shear, t_max, out, hit, n_tris)) return true;                   // its triangles, in order
        cur = pop_next_ref<ANY>(stack, t_max);
    }
    return hit;
}

// One quad-node visit: tests the node's slots, stacks the postponed ones and returns the next reference in `cur`
// (an inner node, a leaf, or REF_NONE when the stack has run dry).
PT_HD uint32_t neg_bits3(const bool neg[3]) { const uint32_t b = (neg[0] ? 1u : 0u) | (neg[1] ? 2u : 0u) | (neg[2] ? 4u : 0u); return b | (b << 8) | (b << 16); } // the ray's sign bits, once per byte (DNode4::pad[0])
template <bool ANY, class Stack, class Geom>
PT_HD void quad_visit(const Geom &G, uint32_t &cur, f3 o, f3 inv, uint32_t px, uint32_t py, uint32_t pz, float t_max, Stack &stack, uint32_t &n_nodes, uint32_t negbits3) {
    // The node keeps its planes per axis: vectors (x lower, x upper, y lower, y upper, z lower, z upper), each with the four slots'
    // values.  Which of a pair the ray meets first is a constant of the ray (px, py, pz: the byte offsets of its near vectors), so the
    // lane reads (near, far) per axis and the four slab tests need no select: the quad visit of round 2 spent 24 v_cndmask_b32 -- a
    // half-rate instruction on gfx950 -- on picking planes after the fetch.
    v4 xn, xf, yn, yf, zn, zf, rf, mt;
    G.quad_load(cur, px, py, pz, xn, xf, yn, yf, zn, zf, rf, mt);
    uint32_t r0 = f2u(rf.x), r1 = f2u(rf.y), r2 = f2u(rf.z), r3 = f2u(rf.w);
    const uint32_t axes = f2u(mt.x);
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    // (an empty slot's box is (+inf, -inf): its test fails by itself)
    bool h0 = slab_planes(xn.x, yn.x, zn.x, xf.x, yf.x, zf.x, o, inv, t0); h0 = h0 & (t0 < t_max);
    bool h1 = slab_planes(xn.y, yn.y, zn.y, xf.y, yf.y, zf.y, o, inv, t1); h1 = h1 & (t1 < t_max);
    bool h2 = slab_planes(xn.z, yn.z, zn.z, xf.z, yf.z, zf.z, o, inv, t2); h2 = h2 & (t2 < t_max);
    bool h3 = slab_planes(xn.w, yn.w, zn.w, xf.w, yf.w, zf.w, o, inv, t3); h3 = h3 & (t3 < t_max);
    n_nodes += (axes >> 12) & 7u;
    const uint32_t sx = negbits3 & f2u(mt.y); // (the split axes as one-hot bytes against the ray's sign bits: = ax < 3 && neg[ax] etc.)
    const bool sw = (sx & 0xffu) != 0u, swa = (sx & 0xff00u) != 0u, swb = (sx & 0xff0000u) != 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64((axes & 0x100u) != 0u) != 0ull) // (rare: the four selects run only in waves that meet such a node)
#endif
    if (axes & 0x100u) { t0 = t1 = t2 = t3 = -3.402823466e38f; } // chunks of one leaf: no pop-time re-test
    // A slot that is not hit loses its reference, so that the reordering below moves two values per slot, not two values and a
    // lane mask (a select between lane masks is computed on 0/1 integers in vector registers); hit <=> reference left.
    r0 = h0 ? r0 : REF_NONE; r1 = h1 ? r1 : REF_NONE; r2 = h2 ? r2 : REF_NONE; r3 = h3 ? r3 : REF_NONE;
#define PT_SWAP(c, ra, ta, rb, tb) { const uint32_t rr = c ? rb : ra; const float tt = c ? tb : ta; rb = c ? ra : rb; tb = c ? ta : tb; ra = rr; ta = tt; }
    PT_SWAP(swa, r0, t0, r1, t1)
    PT_SWAP(swb, r2, t2, r3, t3)
    PT_SWAP(sw, r0, t0, r2, t2)
    PT_SWAP(sw, r1, t1, r3, t3)
#undef PT_SWAP
    h0 = r0 != REF_NONE; h1 = r1 != REF_NONE; h2 = r2 != REF_NONE; h3 = r3 != REF_NONE;
    // visiting order is now 0,1,2,3: the first hit is entered, later hits are stacked last-first
    if (h3 && (h0 || h1 || h2)) stack.push(r3, t3);
    if (h2 && (h0 || h1)) stack.push(r2, t2);
    if (h1 && h0) stack.push(r1, t1);
    if (h0) cur = r0; else if (h1) cur = r1; else if (h2) cur = r2; else if (h3) cur = r3;
    else {
        cur = pop_next_ref<ANY>(stack, t_max);
    }
}

template <bool QUAD, bool ANY, class Stack, class Geom>
PT_HD void node_visit(const Geom &G, uint32_t &cur, f3 o, f3 inv, const bool neg[3], float t_max, Stack &stack, uint32_t &n_nodes, uint32_t negbits3, uint32_t px, uint32_t py, uint32_t pz) {
    if (QUAD) quad_visit<ANY>(G, cur, o, inv, px, py, pz, t_max, stack, n_nodes, negbits3); else pair_visit<ANY>(G, cur, o, inv, neg, t_max, stack, n_nodes);
}
// the byte offsets of a ray's near-plane vectors inside a quad node (the far one of an axis is the other of its pair: offset ^ 16)
PT_HD uint32_t quad_near_x(const bool neg[3]) { return neg[0] ? 16u : 0u; }
PT_HD uint32_t quad_near_y(const bool neg[3]) { return neg[1] ? 48u : 32u; }
PT_HD uint32_t quad_near_z(const bool neg[3]) { return neg[2] ? 80u : 64u; }

// Quad-node traversal (DNode4): one fetch covers two levels of the binary tree.  Slots are visited in the order the
// binary traversal would reach them -- near child's (near, far) grandchildren, then the far child's -- and every
// stacked slot carries its entry distance for the pop-time test.  Testing a grandchild's box directly is equivalent
// to the reference's child-then-grandchild tests because boxes nest exactly (a union of floats is exact) and the slab
// test is monotone in the box: a ray that passes a grandchild's test passes its parent's, with a smaller or equal
// entry distance.
template <bool ANY, bool ALPHA, class Stack, class Geom>
PT_HD bool bvh_trace_quad(const Geom &G, const DScene &sc, f3 o, f3 d, float t_max, Stack &stack, HitRec &out, uint32_t &n_nodes, uint32_t &n_tris) {
    out.prim = -1; out.t = t_max; out.b0 = out.b1 = out.b2 = 0.0f; out.flags = 0;
    stack.clear(); // an any-hit query may have returned early and left entries behind
    if (sc.n_nodes2 + sc.n_nodes4 == 0) return false;
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    const RayShear shear = ray_shear_inv(d, inv);
    uint32_t cur = 0;
    bool hit = false;
    const uint32_t nb3 = neg_bits3(neg);
    while (cur != REF_NONE) {
        while (cur != REF_NONE && !(cur & REF_LEAF)) quad_visit<ANY>(G, cur, o, inv, quad_near_x(neg), quad_near_y(neg), quad_near_z(neg), t_max, stack, n_nodes, nb3);
        if (cur == REF_NONE) break;
        if (leaf_test<ANY, ALPHA>(G, sc, cur, o, shear, t_max, out, hit, n_tris)) return true;
        cur = REF_NONE;
        while (!stack.empty()) { uint32_t r; float te; stack.pop(r, te); if (ANY || te < t_max) { cur = r; break; } }
    }
    return hit;
}

// QUAD selects the node form the geometry source holds (pair nodes: LDS-staged small scenes; quad nodes: the rest)
template <bool QUAD, bool ANY, bool ALPHA, class Stack, class Geom>
PT_HD bool bvh_trace_g(const Geom &G, const DScene &sc, f3 o, f3 d, float t_max, Stack &stack, HitRec &out, uint32_t &n_nodes, uint32_t &n_tris) {
    if (QUAD) return bvh_trace_quad<ANY, ALPHA>(G, sc, o, d, t_max, stack, out, n_nodes, n_tris);
    return bvh_trace_pair<ANY, ALPHA>(G, sc, o, d, t_max, stack, out, n_nodes, n_tris);
}

PT_HD GeomGlobal geom_global(const DScene &sc) {
    GeomGlobal G; G.tris = sc.tris;
    G.nodesv = sc.n_nodes4 ? reinterpret_cast<const v4 *>(sc.nodes4) : reinterpret_cast<const v4 *>(sc.nodes2);
    return G;
}

// Traversal out of global memory in whichever node form the scene carries.
template <bool ANY, bool ALPHA, class Stack>
PT_HD bool bvh_trace_any_form(const DScene &sc, f3 o, f3 d, float t_max, Stack &stack, HitRec &out, uint32_t &n_nodes, uint32_t &n_tris) {
    const GeomGlobal G = geom_global(sc);
    if (sc.n_nodes4) return bvh_trace_quad<ANY, ALPHA>(G, sc, o, d, t_max, stack, out, n_nodes, n_tris);
    return bvh_trace_pair<ANY, ALPHA>(G, sc, o, d, t_max, stack, out, n_nodes, n_tris);
}

template <bool ANY, class Stack>
PT_HD bool bvh_trace(const DScene &sc, f3 o, f3 d, float t_max, Stack &stack, HitRec &out, uint32_t &n_nodes, uint32_t &n_tris) {
    return bvh_trace_any_form<ANY, true>(sc, o, d, t_max, stack, out, n_nodes, n_tris);
}

struct LocalStack { // host twin / small fixed uses
    uint32_t s[128]; float te[128]; int n = 0;
    PT_MEM void push(uint32_t v, float t) { s[n] = v; te[n] = t; ++n; }
    PT_MEM void pop(uint32_t &v, float &t) { --n; v = s[n]; t = te[n]; }
    PT_MEM bool empty() const { return n == 0; }
    PT_MEM void clear() { n = 0; }
};

} // namespace pt
