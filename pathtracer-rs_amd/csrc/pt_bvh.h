// pt_bvh.h -- BVH traversal over the flattened 32-byte-node tree.
//
// Restates BVH::intersect / intersect_p (src/pathtracer/accelerator.rs:359-475) and
// Bounds3::intersect_p_precomp (src/common/bounds.rs:190-232): near-child-first by split axis,
// far child pushed, leaf triangles tested in order (a later hit with equal t replaces the
// earlier one, Q14).  The traversal stack is supplied by the caller: on gfx950 it lives in LDS
// (one column per lane, see kernels.hip), on the host twin it is a local array.
#pragma once
#include "pt_tri.h"

namespace pt {

struct HitRec { int32_t prim; float t, b0, b1, b2; };

PT_HD bool slab_test(const v4 &a, const v4 &b, f3 o, f3 inv, const bool neg[3], float t_max) {
    // a = (pmin.xyz, pmax.x)   b = (pmax.y, pmax.z, offset, meta)
    const float k = 1.0f + 2.0f * gamma_err(3);
    float lox = neg[0] ? a.w : a.x, hix = neg[0] ? a.x : a.w;
    float loy = neg[1] ? b.x : a.y, hiy = neg[1] ? a.y : b.x;
    float loz = neg[2] ? b.y : a.z, hiz = neg[2] ? a.z : b.y;
    float t_min = (lox - o.x) * inv.x;
    float t_mx = (hix - o.x) * inv.x;
    float ty_min = (loy - o.y) * inv.y;
    float ty_max = (hiy - o.y) * inv.y;
    t_mx *= k; ty_max *= k;
    if (t_min > ty_max || ty_min > t_mx) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_mx) t_mx = ty_max;
    float tz_min = (loz - o.z) * inv.z;
    float tz_max = (hiz - o.z) * inv.z;
    tz_max *= k;
    if (t_min > tz_max || tz_min > t_mx) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_mx) t_mx = tz_max;
    return (t_min < t_max) && (t_mx > 0.0f);
}

PT_HD void load_tri(const DTri *tris, uint32_t k, f3 &p0, f3 &p1, f3 &p2, uint32_t &prim, uint32_t &flags) {
    const v4 *q = reinterpret_cast<const v4 *>(tris + k);
    v4 a = q[0], b = q[1], c = q[2];
    p0 = mk3(a.x, a.y, a.z); p1 = mk3(a.w, b.x, b.y); p2 = mk3(b.z, b.w, c.x);
    prim = f2u(c.y); flags = f2u(c.z);
}

// ANY = false: closest hit (intersect); ANY = true: any hit (intersect_p).
// Stack must provide push(uint32_t), pop(), empty() and clear().
//
// "while-while" form: each lane first walks interior nodes until it reaches a leaf that passes the
// slab test (or runs out of nodes), and only then are the leaf's triangles tested -- so that on a
// 64-lane wave the (expensive) triangle phase runs with most lanes active instead of once per
// interior step.  Per ray the sequence of nodes and triangles visited is exactly that of the
// reference loop (accelerator.rs:372-414), only the interleaving between lanes differs.
template <bool ANY, class Stack>
PT_HD bool bvh_trace(const DScene &sc, f3 o, f3 d, float t_max, Stack &stack, HitRec &out, uint32_t &n_nodes, uint32_t &n_tris) {
    out.prim = -1; out.t = t_max; out.b0 = out.b1 = out.b2 = 0.0f;
    stack.clear(); // an any-hit query may have returned early and left entries behind
    if (sc.n_nodes == 0) return false;
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool neg[3] = {inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f};
    const uint32_t NONE = 0xffffffffu;
    uint32_t cur = 0; // node to visit next, NONE when the traversal is over
    bool hit = false;
    while (cur != NONE) {
        // phase 1: descend until a leaf is accepted
        uint32_t leaf_first = 0, leaf_count = 0;
        while (cur != NONE) {
            const v4 *np = reinterpret_cast<const v4 *>(sc.nodes + cur);
            const v4 a = np[0], b = np[1];
            ++n_nodes;
            if (slab_test(a, b, o, inv, neg, t_max)) {
                const uint32_t offset = f2u(b.z), meta = f2u(b.w);
                const uint32_t nprims = meta & 0xffffu;
                if (nprims > 0) {
                    leaf_first = offset; leaf_count = nprims;
                    cur = stack.empty() ? NONE : stack.pop();
                    break;
                }
                const uint32_t axis = (meta >> 16) & 0xffu;
                if (neg[axis]) { stack.push(cur + 1); cur = offset; }
                else { stack.push(offset); cur = cur + 1; }
            } else {
                cur = stack.empty() ? NONE : stack.pop();
            }
        }
        // phase 2: the leaf's triangles, in order
        for (uint32_t i = 0; i < leaf_count; ++i) {
            f3 p0, p1, p2; uint32_t prim, flags;
            load_tri(sc.tris, leaf_first + i, p0, p1, p2, prim, flags);
            ++n_tris;
            TriHit h;
            if (tri_test(o, d, t_max, p0, p1, p2, h) && !(flags & TRI_DEGENERATE)) {
                if (ANY) { out.prim = 0; return true; }
                hit = true; t_max = h.t;
                out.prim = (int32_t)prim; out.t = h.t; out.b0 = h.b0; out.b1 = h.b1; out.b2 = h.b2;
            }
        }
    }
    return hit;
}

struct LocalStack { // host twin / small fixed uses
    uint32_t s[64]; int n = 0;
    PT_MEM void push(uint32_t v) { s[n++] = v; }
    PT_MEM uint32_t pop() { return s[--n]; }
    PT_MEM bool empty() const { return n == 0; }
    PT_MEM void clear() { n = 0; }
};

} // namespace pt
