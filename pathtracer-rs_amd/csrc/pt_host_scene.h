// pt_host_scene.h -- host side of ptrs_scene_create: validates the flat description
// (include/ptrs.h), builds the BVH, and lays the scene out in the device formats of pt_scene.h.
//
// The accelerator is this library's own builder (binned SAH, 12 bins, leaves of <= 4 triangles,
// the same cost model as src/pathtracer/accelerator.rs:156-307 so tree quality is comparable);
// closest-hit results do not depend on the tree.  A caller that already holds the
// reference's flattened tree can pass it through PtrsSceneDesc::bvh_nodes instead.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptrs.h"
#include "pt_light.h"
#include "pt_tri.h"

namespace pt {

struct HostScene {
    std::vector<DNode2> nodes2; // traversal layout, pair form (used when nodes + triangles fit the kernels' LDS staging area)
    std::vector<DNode4> nodes4; // traversal layout, quad form (used otherwise); both derived from `nodes`
    bool use_quad = false;
    std::vector<DNode> nodes;
    std::vector<DTri> tris;
    std::vector<DTriShade> shade;
    std::vector<DMaterial> mats;
    std::vector<DTexture> texs;
    std::vector<DTexLevel> levels;
    std::vector<float> texdata;
    std::vector<DLight> lights;
    std::vector<float> distdata;
    std::vector<uint32_t> inf_lights;
    uint32_t max_depth = 0;
    uint32_t stack_bound = 0; // most entries the pair-node traversal can have stacked (= depth of the pair tree)
    bool kinds_present[7] = {false, false, false, false, false, false, false};
    bool has_alpha = false;
};

namespace hostbvh {
struct Box { float lo[3], hi[3]; };
inline Box empty_box() { Box b; for (int k = 0; k < 3; ++k) { b.lo[k] = 3.402823466e38f; b.hi[k] = -3.402823466e38f; } return b; }
inline void grow(Box &b, const Box &o) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
inline void grow_pt(Box &b, const float *p) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], p[k]); b.hi[k] = std::max(b.hi[k], p[k]); } }
inline float area(const Box &b) { float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2]; return 2.0f * (dx * dy + dx * dz + dy * dz); }

struct Item { uint32_t prim; float c[3]; Box b; };

// recursive top-down build emitting depth-first nodes (first child = parent + 1)
struct Builder {
    std::vector<Item> items;
    std::vector<DNode> *nodes;
    std::vector<uint32_t> order;
    uint32_t max_depth = 0;
    bool overflow = false;
    int max_leaf = 4; bool force_leaf = false; // leaf policy; measured on Cornell: SAH leaves (17.0 nodes, 1.9 triangles per ray) beat forced fat leaves

    uint32_t build(size_t lo, size_t hi, uint32_t depth) {
        uint32_t me = (uint32_t)nodes->size();
        nodes->push_back(DNode());
        max_depth = std::max(max_depth, depth);
        Box bounds = empty_box(), cb = empty_box();
        for (size_t i = lo; i < hi; ++i) { grow(bounds, items[i].b); grow_pt(cb, items[i].c); }
        const size_t n = hi - lo;
        int axis = 0;
        { float ex[3] = {cb.hi[0] - cb.lo[0], cb.hi[1] - cb.lo[1], cb.hi[2] - cb.lo[2]};
          if (ex[1] > ex[axis]) axis = 1;
          if (ex[2] > ex[axis]) axis = 2; }
        size_t mid = lo;
        bool leaf = (n == 1) || (cb.hi[axis] == cb.lo[axis]);
        if (!leaf) {
            if (n <= 2) {
                mid = (lo + hi) / 2;
                std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, [axis](const Item &a, const Item &b) { return a.c[axis] < b.c[axis]; });
            } else {
                const int NB = 12;
                size_t cnt[NB] = {0}; Box bb[NB];
                for (int k = 0; k < NB; ++k) bb[k] = empty_box();
                const float inv = 1.0f / (cb.hi[axis] - cb.lo[axis]);
                auto bin_of = [&](const Item &it) { int b = (int)((float)NB * ((it.c[axis] - cb.lo[axis]) * inv)); return b >= NB ? NB - 1 : (b < 0 ? 0 : b); };
                for (size_t i = lo; i < hi; ++i) { int b = bin_of(items[i]); cnt[b]++; grow(bb[b], items[i].b); }
                // sweep from both ends
                float left_area[NB - 1], right_area[NB - 1]; size_t left_cnt[NB - 1], right_cnt[NB - 1];
                { Box acc = empty_box(); size_t c = 0; for (int k = 0; k < NB - 1; ++k) { grow(acc, bb[k]); c += cnt[k]; left_area[k] = c ? area(acc) : 0.0f; left_cnt[k] = c; } }
                { Box acc = empty_box(); size_t c = 0; for (int k = NB - 1; k >= 1; --k) { grow(acc, bb[k]); c += cnt[k]; right_area[k - 1] = c ? area(acc) : 0.0f; right_cnt[k - 1] = c; } }
                const float inv_area = 1.0f / area(bounds);
                int best = 0; float best_cost = 3.402823466e38f;
                for (int k = 0; k < NB - 1; ++k) {
                    float cst = 1.0f + ((float)left_cnt[k] * left_area[k] + (float)right_cnt[k] * right_area[k]) * inv_area;
                    if (cst < best_cost) { best_cost = cst; best = k; }
                }
                if (n > (size_t)max_leaf || (!force_leaf && best_cost < (float)n)) {
                    auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item &x) { return bin_of(x) <= best; });
                    mid = (size_t)(it - items.begin());
                    if (mid == lo || mid == hi) { // numerically empty side: fall back to a median split
                        mid = (lo + hi) / 2;
                        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, [axis](const Item &a, const Item &b) { return a.c[axis] < b.c[axis]; });
                    }
                } else leaf = true;
            }
        }
        DNode nd;
        nd.pmin[0] = bounds.lo[0]; nd.pmin[1] = bounds.lo[1]; nd.pmin[2] = bounds.lo[2];
        nd.pmax0 = bounds.hi[0]; nd.pmax1 = bounds.hi[1]; nd.pmax2 = bounds.hi[2];
        if (leaf) {
            nd.offset = (uint32_t)order.size();
            if (n > 0xffffu) overflow = true; // only possible when > 65535 centroids coincide
            nd.meta = (uint32_t)(n & 0xffffu);
            for (size_t i = lo; i < hi; ++i) order.push_back(items[i].prim);
            (*nodes)[me] = nd;
            return me;
        }
        build(lo, mid, depth + 1);
        uint32_t second = build(mid, hi, depth + 1);
        nd.offset = second; nd.meta = (uint32_t)axis << 16;
        (*nodes)[me] = nd;
        return me;
    }
};
} // namespace hostbvh

// node_form: 0 = by size (pair nodes when the tree fits the LDS staging area, quad nodes otherwise), 1 = pair, 2 = quad (test hook)
// node_order (quad form): 0 = the tree's top QUAD_TOP_NODES records breadth-first (what the traversal kernels cache in LDS), the rest in the
// builder's depth-first order; 1 = treelets: behind the top, every subtree of three quad levels (1 + 4 + 16 records = 2.7 KB) sits in
// consecutive records, so that a ray's next two fetches after entering a treelet fall into lines its neighbours in the wave are
// fetching too.  An order, not a different tree: same visits, same results.
inline int build_host_scene(const PtrsSceneDesc &d, HostScene &H, std::string &err, int node_form = 0, int node_order = 0) {
    auto bad = [&](const char *m) { err = m; return (int)PTRS_ERR_INVALID; };
    if (d.n_meshes && !d.meshes) return bad("meshes is NULL");
    // ---- textures -------------------------------------------------------------------------------
    H.texs.resize(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; ++i) {
        const PtrsTexture &s = d.textures[i]; DTexture &t = H.texs[i];
        std::memset(&t, 0, sizeof(t));
        if (s.kind < 0 || s.kind > 2 || (s.channels != 1 && s.channels != 3)) return bad("bad texture record");
        t.kind = s.kind; t.channels = s.channels;
        for (int c = 0; c < 3; ++c) { t.value[c] = s.value[c]; t.value2[c] = s.value2[c]; }
        t.su = s.su; t.sv = s.sv; t.du = s.du; t.dv = s.dv; t.wrap = s.wrap;
        if (s.kind == PTRS_TEX_IMAGE) {
            if (s.n_levels <= 0 || !s.level_data || !s.level_cols || !s.level_rows) return bad("image texture without pyramid");
            t.n_levels = s.n_levels; t.first_level = (uint32_t)H.levels.size();
            for (int l = 0; l < s.n_levels; ++l) {
                DTexLevel L; L.offset = H.texdata.size(); L.cols = s.level_cols[l]; L.rows = s.level_rows[l];
                if (L.cols <= 0 || L.rows <= 0) return bad("empty texture level");
                size_t n = (size_t)L.cols * (size_t)L.rows * (size_t)s.channels;
                H.texdata.insert(H.texdata.end(), s.level_data[l], s.level_data[l] + n);
                H.levels.push_back(L);
            }
        }
    }
    auto tex_ok = [&](int32_t id, int ch) { return id >= 0 && (uint32_t)id < d.n_textures && d.textures[id].channels == ch; };
    // ---- materials ------------------------------------------------------------------------------
    H.mats.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; ++i) {
        const PtrsMaterial &s = d.materials[i]; DMaterial &m = H.mats[i];
        std::memset(&m, 0, sizeof(m));
        m.kind = s.kind; m.flags = s.flags; m.inner = s.inner;
        for (int k = 0; k < 6; ++k) {
            m.tex[k] = s.tex[k];
            if (s.tex[k] >= 0 && (uint32_t)s.tex[k] < d.n_textures && d.textures[s.tex[k]].kind == PTRS_TEX_CONSTANT) { // fold ConstantTexture values
                m.const_mask |= 1u << k;
                for (int c = 0; c < 3; ++c) m.cval[k][c] = d.textures[s.tex[k]].value[c];
            }
        }
        bool ok = true;
        for (int k = 0; k < 6; ++k) if (s.tex[k] < -1 || (s.tex[k] >= 0 && (uint32_t)s.tex[k] >= d.n_textures)) return bad("material texture id out of range"); // every slot is -1 or a texture, also the ones its kind ignores
        switch (s.kind) {
            case PTRS_MAT_MATTE: ok = tex_ok(s.tex[0], 3); break;
            case PTRS_MAT_MIRROR: break;
            case PTRS_MAT_GLASS: ok = tex_ok(s.tex[0], 3) && tex_ok(s.tex[1], 3) && tex_ok(s.tex[2], 1); break;
            case PTRS_MAT_METAL: // u/v roughness: both or neither (pt_material.h prefers them slot by slot); `roughness` is needed for the slots they leave open
                ok = tex_ok(s.tex[0], 3) && tex_ok(s.tex[1], 3) && tex_ok(s.tex[2], 3) && ((tex_ok(s.tex[4], 1) && tex_ok(s.tex[5], 1)) || (s.tex[4] < 0 && s.tex[5] < 0 && tex_ok(s.tex[3], 1))) && (s.tex[3] < 0 || tex_ok(s.tex[3], 1));
                break;
            case PTRS_MAT_DISNEY: ok = tex_ok(s.tex[0], 3) && tex_ok(s.tex[1], 1) && tex_ok(s.tex[2], 1) && tex_ok(s.tex[3], 1); break;
            case PTRS_MAT_SUBSTRATE: ok = tex_ok(s.tex[0], 3) && tex_ok(s.tex[1], 3) && tex_ok(s.tex[2], 1) && tex_ok(s.tex[3], 1); break;
            case PTRS_MAT_NORMAL: ok = tex_ok(s.tex[0], 3) && s.inner >= 0 && (uint32_t)s.inner < d.n_materials && (uint32_t)s.inner != i; break;
            default: err = "unknown material kind"; return PTRS_ERR_UNSUPPORTED;
        }
        if (!ok) return bad("material references a missing texture / inner material");
    }
    // ---- triangles ------------------------------------------------------------------------------
    std::vector<uint32_t> mesh_first(d.n_meshes);
    size_t n_tris = 0;
    for (uint32_t m = 0; m < d.n_meshes; ++m) { mesh_first[m] = (uint32_t)n_tris; n_tris += d.meshes[m].n_tris; }
    if (n_tris >= 0x7fffffffull) return bad("too many triangles");
    H.shade.resize(n_tris);
    for (uint32_t m = 0; m < d.n_meshes; ++m) {
        const PtrsMesh &s = d.meshes[m];
        if (!s.pos || !s.indices) return bad("mesh without positions / indices");
        if (s.material < 0 || (uint32_t)s.material >= d.n_materials) return bad("mesh material out of range");
        if (s.alpha_mask_tex >= 0 && !tex_ok(s.alpha_mask_tex, 1)) return bad("alpha mask must be a 1-channel texture");
        if (s.alpha_mask_tex >= 0) H.has_alpha = true;
        int mi = s.material;
        for (int g = 0; g < 4 && H.mats[mi].kind == PTRS_MAT_NORMAL; ++g) mi = H.mats[mi].inner;
        const int bucket = H.mats[mi].kind;
        if (bucket == PTRS_MAT_NORMAL) return bad("normal materials nested too deeply");
        H.kinds_present[bucket] = true;
        for (uint32_t t = 0; t < s.n_tris; ++t) {
            DTriShade &T = H.shade[mesh_first[m] + t];
            std::memset(&T, 0, sizeof(T));
            uint32_t v[3] = {s.indices[3 * t], s.indices[3 * t + 1], s.indices[3 * t + 2]};
            for (int k = 0; k < 3; ++k) if (v[k] >= s.n_verts) return bad("vertex index out of range");
            auto P3 = [&](const float *src, uint32_t vi, int c) { return src[3 * vi + c]; };
            T.p0[0] = P3(s.pos, v[0], 0); T.p0[1] = P3(s.pos, v[0], 1); T.p0[2] = P3(s.pos, v[0], 2);
            T.p1x = P3(s.pos, v[1], 0); T.p1y = P3(s.pos, v[1], 1); T.p1z = P3(s.pos, v[1], 2);
            T.p2x = P3(s.pos, v[2], 0); T.p2y = P3(s.pos, v[2], 1); T.p2z = P3(s.pos, v[2], 2);
            if (s.normal) {
                T.n0[0] = P3(s.normal, v[0], 0); T.n0[1] = P3(s.normal, v[0], 1); T.n0[2] = P3(s.normal, v[0], 2);
                T.n1x = P3(s.normal, v[1], 0); T.n1y = P3(s.normal, v[1], 1); T.n1z = P3(s.normal, v[1], 2);
                T.n2x = P3(s.normal, v[2], 0); T.n2y = P3(s.normal, v[2], 1); T.n2z = P3(s.normal, v[2], 2);
            }
            if (s.tangent) {
                T.s0[0] = P3(s.tangent, v[0], 0); T.s0[1] = P3(s.tangent, v[0], 1); T.s0[2] = P3(s.tangent, v[0], 2);
                T.s1x = P3(s.tangent, v[1], 0); T.s1y = P3(s.tangent, v[1], 1); T.s1z = P3(s.tangent, v[1], 2);
                T.s2x = P3(s.tangent, v[2], 0); T.s2y = P3(s.tangent, v[2], 1); T.s2z = P3(s.tangent, v[2], 2);
            }
            const float duv[3][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {1.0f, 1.0f}}; // shape.rs:41-46
            float *UV[3] = {T.uv0, T.uv1, T.uv2};
            for (int k = 0; k < 3; ++k) for (int c = 0; c < 2; ++c) UV[k][c] = s.uv ? s.uv[2 * v[k] + c] : duv[k][c];
            T.material = s.material; T.light = -1; T.alpha_tex = s.alpha_mask_tex;
            T.flags = (s.normal ? TRI_HAS_NORMAL : 0u) | (s.tangent ? TRI_HAS_TANGENT : 0u) | (s.reverse_orientation ? TRI_REVERSE : 0u) | (s.transform_swaps_handedness ? TRI_SWAPS : 0u) |
                      (s.alpha_mask_tex >= 0 ? TRI_HAS_ALPHA : 0u) | ((uint32_t)bucket << TRI_BUCKET_SHIFT);
            f3 dpdu, dpdv, ng, ssn;
            const f3 q0 = mk3(T.p0[0], T.p0[1], T.p0[2]), q1 = mk3(T.p1x, T.p1y, T.p1z), q2 = mk3(T.p2x, T.p2y, T.p2z);
            const f2 t0 = mk2(T.uv0[0], T.uv0[1]), t1 = mk2(T.uv1[0], T.uv1[1]), t2 = mk2(T.uv2[0], T.uv2[1]);
            if (!tri_dpduv(q0, q1, q2, t0, t1, t2, dpdu, dpdv)) T.flags |= TRI_DEGENERATE;
            tri_constants(q0, q1, q2, t0, t1, t2, T.flags, ng, ssn, dpdu, dpdv); // what every hit on this triangle would recompute (shape.rs:187-266)
            T.ng[0] = ng.x; T.ng[1] = ng.y; T.ng[2] = ng.z; T.ssn0 = ssn.x; T.ssn1 = ssn.y; T.ssn2 = ssn.z;
            T.dpdu0 = dpdu.x; T.dpdu1 = dpdu.y; T.dpdu2 = dpdu.z; T.dpdv[0] = dpdv.x; T.dpdv[1] = dpdv.y; T.dpdv[2] = dpdv.z;
        }
    }
    // ---- lights ---------------------------------------------------------------------------------
    H.lights.resize(d.n_lights);
    for (uint32_t i = 0; i < d.n_lights; ++i) {
        const PtrsLight &s = d.lights[i]; DLight &L = H.lights[i];
        std::memset(&L, 0, sizeof(L));
        L.kind = s.kind; L.tri = -1; L.ke_tex = -1; L.lmap_tex = -1;
        for (int c = 0; c < 3; ++c) { L.v[c] = s.v[c]; L.c[c] = s.c[c]; }
        L.world_radius = s.world_radius;
        if (s.kind == PTRS_LIGHT_AREA) {
            if (s.mesh >= d.n_meshes || s.tri >= d.meshes[s.mesh].n_tris || !tex_ok(s.ke_tex, 3)) return bad("bad area light record");
            L.tri = (int32_t)(mesh_first[s.mesh] + s.tri); L.ke_tex = s.ke_tex;
            DTriShade &T = H.shade[L.tri];
            T.light = (int32_t)i; T.flags |= TRI_IS_LIGHT;
            const f3 q0 = mk3(T.p0[0], T.p0[1], T.p0[2]), q1 = mk3(T.p1x, T.p1y, T.p1z), q2 = mk3(T.p2x, T.p2y, T.p2z);
            L.area = 0.5f * len(cross(q1 - q0, q2 - q0)); // Triangle::area, shape.rs:533-539
            if (d.textures[s.ke_tex].kind == PTRS_TEX_CONSTANT) { L.ke_const = 1u; for (int c = 0; c < 3; ++c) L.c[c] = d.textures[s.ke_tex].value[c]; }
        } else if (s.kind == PTRS_LIGHT_INFINITE) {
            if (!tex_ok(s.lmap_tex, 3) || d.textures[s.lmap_tex].kind != PTRS_TEX_IMAGE || s.dist_nu <= 0 || s.dist_nv <= 0 || !s.dist_func || !s.dist_cdf || !s.dist_func_int || !s.marg_cdf) return bad("bad infinite light record");
            L.lmap_tex = s.lmap_tex;
            std::memcpy(L.l2w, s.light_to_world, 48); std::memcpy(L.w2l, s.world_to_light, 48);
            L.nu = s.dist_nu; L.nv = s.dist_nv; L.marg_int = s.marg_func_int;
            size_t nu = (size_t)s.dist_nu, nv = (size_t)s.dist_nv;
            L.func_off = (uint32_t)H.distdata.size(); H.distdata.insert(H.distdata.end(), s.dist_func, s.dist_func + nu * nv);
            L.cdf_off = (uint32_t)H.distdata.size(); H.distdata.insert(H.distdata.end(), s.dist_cdf, s.dist_cdf + (nu + 1) * nv);
            L.fint_off = (uint32_t)H.distdata.size(); H.distdata.insert(H.distdata.end(), s.dist_func_int, s.dist_func_int + nv);
            L.mcdf_off = (uint32_t)H.distdata.size(); H.distdata.insert(H.distdata.end(), s.marg_cdf, s.marg_cdf + nv + 1);
            // guide tables (pt_vec.h find_interval_cdf_guided): g = the table size rounded up to a power of two
            auto add_guide = [&](const float *cdf, size_t n, uint32_t g) { // cdf has n+1 entries
                size_t cnt = 0;
                for (uint32_t k = 0; k <= g; ++k) {
                    const float x = (float)k / (float)g;
                    while (cnt < n + 1 && cdf[cnt] <= x) ++cnt;
                    H.distdata.push_back(ptf_from_bits((uint32_t)cnt));
                }
            };
            uint32_t gu = 1, gv = 1; while (gu < nu) gu <<= 1; while (gv < nv) gv <<= 1;
            bool monotone = true; // the shortcut needs what Distribution1D::new produces anyway
            for (size_t r = 0; r < nv && monotone; ++r) for (size_t k = 0; k < nu; ++k) if (!(s.dist_cdf[r * (nu + 1) + k] <= s.dist_cdf[r * (nu + 1) + k + 1])) { monotone = false; break; }
            for (size_t k = 0; k < nv && monotone; ++k) if (!(s.marg_cdf[k] <= s.marg_cdf[k + 1])) monotone = false;
            if (monotone && H.distdata.size() + (size_t)(gu + 1) * nv + gv + 1 < 0xffffffffull) {
                L.guide_u = gu; L.guide_v = gv;
                L.mguide_off = (uint32_t)H.distdata.size(); add_guide(s.marg_cdf, nv, gv);
                L.cguide_off = (uint32_t)H.distdata.size();
                for (size_t r = 0; r < nv; ++r) add_guide(s.dist_cdf + r * (nu + 1), nu, gu);
            }
            H.inf_lights.push_back(i);
        } else if (s.kind != PTRS_LIGHT_POINT && s.kind != PTRS_LIGHT_DIRECTIONAL) { err = "unknown light kind"; return PTRS_ERR_UNSUPPORTED; }
    }
    for (auto &L : H.lights) if (L.kind == PTRS_LIGHT_AREA) {
        L.T = H.shade[L.tri];
        // Triangle::sample and pdf_at_point end with a face-forwarded geometric normal.  When every vertex normal lies
        // clearly on one side of the triangle's plane the sign cannot depend on the barycentrics: run the device code
        // at seven points, require identical bits, and let the kernels use the constant (pt_light.h).
        const TriRegs T = load_tri_regs(&L.T);
        bool ok = !(T.flags & TRI_DEGENERATE);
        if (ok && (T.flags & TRI_HAS_NORMAL)) {
            const f3 g = normalize(cross(T.p0 - T.p2, T.p1 - T.p2));
            const float d0 = dot(g, T.n0) / len(T.n0), d1 = dot(g, T.n1) / len(T.n1), d2 = dot(g, T.n2) / len(T.n2);
            ok = (d0 > 0.05f && d1 > 0.05f && d2 > 0.05f) || (d0 < -0.05f && d1 < -0.05f && d2 < -0.05f);
        }
        if (ok) {
            static const float B[7][2] = {{1, 0}, {0, 1}, {0, 0}, {0.5f, 0.5f}, {0.5f, 0}, {0, 0.5f}, {0.3333f, 0.3333f}};
            f3 ns0 = splat3(0.0f), np0 = splat3(0.0f);
            for (int k = 0; k < 7 && ok; ++k) {
                f3 p, n, perr; f2 uv;
                tri_point_normal(T, B[k][0], B[k][1], 1.0f - B[k][0] - B[k][1], p, n);
                const float su0 = 1.0f - B[k][0]; // b0 = 1 - sqrt(u0), b1 = u1 sqrt(u0)
                f3 n2; tri_sample(T, mk2(su0 * su0, su0 > 0.0f ? B[k][1] / su0 : 0.0f), p, n2, perr, uv);
                if (k == 0) { np0 = n; ns0 = n2; }
                else ok = f2u(n.x) == f2u(np0.x) && f2u(n.y) == f2u(np0.y) && f2u(n.z) == f2u(np0.z) && f2u(n2.x) == f2u(ns0.x) && f2u(n2.y) == f2u(ns0.y) && f2u(n2.z) == f2u(ns0.z);
            }
            if (ok && np0.x == np0.x && ns0.x == ns0.x) { L.n_ok = 1u; L.n_point[0] = np0.x; L.n_point[1] = np0.y; L.n_point[2] = np0.z; L.n_sample[0] = ns0.x; L.n_sample[1] = ns0.y; L.n_sample[2] = ns0.z; }
        }
    }
    // ---- accelerator ----------------------------------------------------------------------------
    std::vector<uint32_t> order;
    if (d.bvh_nodes && d.n_bvh_nodes) {
        if (!d.bvh_prims) return bad("bvh_nodes without bvh_prims");
        static_assert(sizeof(PtrsBvhNode) == sizeof(DNode), "node layouts must agree");
        H.nodes.resize(d.n_bvh_nodes);
        std::memcpy(H.nodes.data(), d.bvh_nodes, sizeof(DNode) * d.n_bvh_nodes);
        order.assign(d.bvh_prims, d.bvh_prims + n_tris);
        for (auto &nd : H.nodes) { // validate indices so that a bad tree cannot fault the GPU
            uint32_t np = nd.meta & 0xffffu;
            if (np ? (nd.offset + np > n_tris) : (nd.offset >= d.n_bvh_nodes)) return bad("bvh node out of range");
            if (!np && ((nd.meta >> 16) & 0xffu) > 2) return bad("bvh axis out of range");
        }
        for (uint32_t p : order) if (p >= n_tris) return bad("bvh prim out of range");
        // depth by walking the depth-first layout
        std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0, 1});
        while (!st.empty()) { auto [i, dp] = st.back(); st.pop_back(); H.max_depth = std::max(H.max_depth, dp); const DNode &nd = H.nodes[i]; if (!(nd.meta & 0xffffu)) { if (i + 1 >= d.n_bvh_nodes || dp > 4096) return bad("bvh malformed"); st.push_back({i + 1, dp + 1}); st.push_back({nd.offset, dp + 1}); } }
    } else if (n_tris) {
        hostbvh::Builder B; B.nodes = &H.nodes; B.items.resize(n_tris);
        for (size_t i = 0; i < n_tris; ++i) {
            hostbvh::Item &it = B.items[i]; it.prim = (uint32_t)i; it.b = hostbvh::empty_box();
            const DTriShade &T = H.shade[i];
            const float q1[3] = {T.p1x, T.p1y, T.p1z}, q2[3] = {T.p2x, T.p2y, T.p2z};
            hostbvh::grow_pt(it.b, T.p0); hostbvh::grow_pt(it.b, q1); hostbvh::grow_pt(it.b, q2);
            for (int k = 0; k < 3; ++k) it.c[k] = it.b.lo[k] + 0.5f * (it.b.hi[k] - it.b.lo[k]);
        }
        H.nodes.reserve(2 * n_tris);
        B.build(0, n_tris, 1);
        if (B.overflow) return bad("leaf with more than 65535 coincident triangles");
        order.swap(B.order);
        H.max_depth = B.max_depth;
        for (auto &nd : H.nodes) if ((nd.meta & 0xffffu) == 0 && (nd.meta >> 16) > 2) return bad("internal: bad axis");
    }
    uint32_t stack_bound2 = 0, stack_bound4 = 0;
    // ---- quad nodes for traversal (pt_scene.h DNode4): two binary levels per record ----
    if (!H.nodes.empty()) {
        if (n_tris > REF_FIRST_MASK) return bad("too many triangles for the leaf reference encoding");
        struct Conv {
            const std::vector<DNode> &n; std::vector<DNode4> &out;
            static void set_slot(DNode4 &d, int s, const DNode &nd, uint32_t ref) {
                const float lo[3] = {nd.pmin[0], nd.pmin[1], nd.pmin[2]}, hi[3] = {nd.pmax0, nd.pmax1, nd.pmax2};
                for (int a = 0; a < 3; ++a) { d.box[(2 * a) * 4 + s] = lo[a]; d.box[(2 * a + 1) * 4 + s] = hi[a]; } // per axis: the four slots' lower planes, then their upper planes
                d.ref[s] = ref;
            }
            // an empty slot holds the box no ray can enter (min = +inf, max = -inf: its slab test ends with t_max_box = -inf > 0 false for every
            // finite origin and any direction, infinite reciprocals included), so traversal needs no test for REF_NONE
            static DNode4 blank() { DNode4 d; std::memset(&d, 0, sizeof(d)); for (int s = 0; s < 4; ++s) { d.ref[s] = REF_NONE; for (int k = 0; k < 3; ++k) { d.box[(2 * k) * 4 + s] = PT_INF; d.box[(2 * k + 1) * 4 + s] = -PT_INF; } } return d; }
            // reference to a leaf range; ranges longer than REF_MAX_LEAF (only possible when many centroids coincide)
            // become quad nodes whose slots are consecutive chunks sharing the leaf's box, visited in order and
            // popped without the entry re-test (the reference tests such a leaf's box once)
            uint32_t leaf_ref(const DNode &leaf, uint32_t first, uint32_t count) {
                if (count <= REF_MAX_LEAF) return REF_LEAF | ((count - 1u) << REF_COUNT_SHIFT) | first;
                const uint32_t me = (uint32_t)out.size();
                out.push_back(blank());
                DNode4 d = blank(); d.axes = 3u | (3u << 2) | (3u << 4) | 0x100u;
                const uint32_t per = (count + 3u) / 4u;
                uint32_t at = first, left = count;
                for (int s = 0; s < 4 && left; ++s) { const uint32_t c = std::min(per, left); set_slot(d, s, leaf, leaf_ref(leaf, at, c)); at += c; left -= c; }
                out[me] = d;
                return me;
            }
            uint32_t ref_of(uint32_t i) { // reference for binary-tree node i
                const DNode &nd = n[i];
                const uint32_t np = nd.meta & 0xffffu;
                if (np) return leaf_ref(nd, nd.offset, np);
                const uint32_t me = (uint32_t)out.size();
                out.push_back(blank());
                DNode4 d = blank();
                const uint32_t kid[2] = {i + 1, nd.offset};
                uint32_t ax[2] = {3u, 3u};
                for (int g = 0; g < 2; ++g) {
                    const DNode &c = n[kid[g]];
                    if (c.meta & 0xffffu) set_slot(d, 2 * g, c, leaf_ref(c, c.offset, c.meta & 0xffffu));
                    else { ax[g] = (c.meta >> 16) & 0xffu; set_slot(d, 2 * g, n[kid[g] + 1], ref_of(kid[g] + 1)); set_slot(d, 2 * g + 1, n[c.offset], ref_of(c.offset)); }
                }
                d.axes = ((nd.meta >> 16) & 3u) | (ax[0] << 2) | (ax[1] << 4);
                out[me] = d;
                return me;
            }
        } conv{H.nodes, H.nodes4};
        H.nodes4.reserve(H.nodes.size() / 3 + 2);
        const uint32_t root = conv.ref_of(0);
        if (root & REF_LEAF) { // the whole scene is one leaf: a node whose only slot is that leaf
            DNode4 d = Conv::blank(); Conv::set_slot(d, 0, H.nodes[0], root); d.axes = 3u | (3u << 2) | (3u << 4);
            H.nodes4.insert(H.nodes4.begin(), d);
        }
        if (H.nodes4.size() >= (1u << 25)) return bad("too many quad nodes for 32-bit node offsets (2^25 x 128 bytes)"); // GeomGlobal::quad_load
        for (DNode4 &d : H.nodes4) { uint32_t c = 0; for (int sl = 0; sl < 4; ++sl) c += d.ref[sl] != REF_NONE ? 1u : 0u; d.axes = (d.axes & 0xfffu) | (c << 12); } // bits 12-14: occupied slots (the boxes-tested statistic)
        // pad[0]: the three split axes once more as one-hot bytes (bit a of byte k set when split k runs along axis a, 0 = never swap): with the
        // ray's sign bits replicated into three bytes, `swap split k` is one AND and one compare instead of an indexed read of the sign array
        for (DNode4 &d : H.nodes4) { uint32_t m = 0; for (int k = 0; k < 3; ++k) { const uint32_t a = (d.axes >> (2 * k)) & 3u; if (a < 3u) m |= (1u << a) << (8 * k); } d.pad[0] = m; }
        // stack bound: at most three entries are stacked per level of the quad tree
        std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0u, 1u});
        uint32_t depth4 = 0;
        while (!st.empty()) {
            auto [i, dp] = st.back(); st.pop_back();
            depth4 = std::max(depth4, dp);
            const DNode4 &nd = H.nodes4[i];
            for (int s2 = 0; s2 < 4; ++s2) if (nd.ref[s2] != REF_NONE && !(nd.ref[s2] & REF_LEAF)) st.push_back({nd.ref[s2], dp + 1});
        }
        H.max_depth = std::max(H.max_depth, depth4 + 1);
        stack_bound4 = 3u * depth4;
    }
    // ---- pair nodes for traversal (pt_scene.h DNode2) ----
    if (!H.nodes.empty()) {
        struct Conv {
            const std::vector<DNode> &n; std::vector<DNode2> &out;
            static void box_of(const DNode &nd, float mn[3], float mx[3]) { mn[0] = nd.pmin[0]; mn[1] = nd.pmin[1]; mn[2] = nd.pmin[2]; mx[0] = nd.pmax0; mx[1] = nd.pmax1; mx[2] = nd.pmax2; }
            static void set_child(DNode2 &d, int which, const float mn[3], const float mx[3], uint32_t ref) {
                if (which == 0) { d.c0min[0] = mn[0]; d.c0min[1] = mn[1]; d.c0min[2] = mn[2]; d.c0max0 = mx[0]; d.c0max1 = mx[1]; d.c0max2 = mx[2]; d.ref0 = ref; }
                else { d.c1min0 = mn[0]; d.c1min1 = mn[1]; d.c1min2 = mn[2]; d.c1max[0] = mx[0]; d.c1max[1] = mx[1]; d.c1max[2] = mx[2]; d.ref1 = ref; }
            }
            // reference to a leaf range; ranges longer than REF_MAX_LEAF (only possible when many centroids
            // coincide) become a balanced subtree of pair nodes that share the leaf's box and keep the triangle order
            uint32_t leaf_ref(uint32_t first, uint32_t count, const float mn[3], const float mx[3]) {
                if (count <= REF_MAX_LEAF) return REF_LEAF | ((count - 1u) << REF_COUNT_SHIFT) | first;
                uint32_t me = (uint32_t)out.size();
                out.push_back(DNode2());
                DNode2 d; std::memset(&d, 0, sizeof(d));
                const uint32_t half = count / 2u; // balanced, order-preserving split
                uint32_t r0 = leaf_ref(first, half, mn, mx);
                uint32_t r1 = leaf_ref(first + half, count - half, mn, mx);
                set_child(d, 0, mn, mx, r0); set_child(d, 1, mn, mx, r1); d.axis = 3; // axis 3: never "negative" -> first child first
                out[me] = d;
                return me;
            }
            uint32_t ref_of(uint32_t i) { // reference for binary-tree node i
                const DNode &nd = n[i];
                float mn[3], mx[3]; box_of(nd, mn, mx);
                const uint32_t np = nd.meta & 0xffffu;
                if (np) return leaf_ref(nd.offset, np, mn, mx);
                uint32_t me = (uint32_t)out.size();
                out.push_back(DNode2());
                DNode2 d; std::memset(&d, 0, sizeof(d));
                float amn[3], amx[3], bmn[3], bmx[3];
                box_of(n[i + 1], amn, amx); box_of(n[nd.offset], bmn, bmx);
                const uint32_t ra = ref_of(i + 1), rb = ref_of(nd.offset);
                set_child(d, 0, amn, amx, ra); set_child(d, 1, bmn, bmx, rb); d.axis = (nd.meta >> 16) & 0xffu;
                out[me] = d;
                return me;
            }
        } conv{H.nodes, H.nodes2};
        if (n_tris > REF_FIRST_MASK) return bad("too many triangles for the leaf reference encoding");
        H.nodes2.reserve(H.nodes.size() / 2 + 2);
        const uint32_t root = conv.ref_of(0);
        if (root & REF_LEAF) { // the whole scene is one leaf: give it a parent whose second child is absent
            DNode2 d; std::memset(&d, 0, sizeof(d));
            float mn[3], mx[3]; Conv::box_of(H.nodes[0], mn, mx);
            Conv::set_child(d, 0, mn, mx, root); d.ref1 = REF_NONE; d.axis = 3;
            H.nodes2.insert(H.nodes2.begin(), d);
        }
        // stack bound = depth of the pair-node tree
        std::vector<std::pair<uint32_t, uint32_t>> st; st.push_back({0u, 1u});
        uint32_t depth2 = 0;
        while (!st.empty()) {
            auto [i, dp] = st.back(); st.pop_back();
            depth2 = std::max(depth2, dp);
            const DNode2 &nd = H.nodes2[i];
            if (!(nd.ref0 & REF_LEAF)) st.push_back({nd.ref0, dp + 1});
            if (nd.ref1 != REF_NONE && !(nd.ref1 & REF_LEAF)) st.push_back({nd.ref1, dp + 1});
        }
        H.max_depth = std::max(H.max_depth, depth2 + 1);
        stack_bound2 = depth2;
    }
    // small scenes are walked out of LDS in pair form (cheaper steps when a fetch costs nothing), the rest in quad form
    H.use_quad = 7u * H.nodes2.size() + 9u * n_tris > PAIR_FORM_MAX_V4;
    if (node_form == 2) H.use_quad = true; // (pair form cannot be forced: the kernels only walk it out of LDS)
    if (H.use_quad) { H.nodes2.clear(); H.stack_bound = stack_bound4; } else { H.nodes4.clear(); H.stack_bound = stack_bound2; }
    if (H.use_quad && H.nodes4.size() > QUAD_TOP_NODES) {
        // breadth-first order for the top of the tree: the first QUAD_TOP_NODES records are the ones every ray starts in,
        // and the traversal kernels keep them in LDS
        std::vector<uint32_t> newid(H.nodes4.size(), 0xffffffffu), seq; seq.reserve(H.nodes4.size());
        auto kids = [&](uint32_t o, std::vector<uint32_t> &out) { for (int sl = 0; sl < 4; ++sl) { const uint32_t r = H.nodes4[o].ref[sl]; if (r != REF_NONE && !(r & REF_LEAF)) out.push_back(r); } };
        std::vector<uint32_t> roots; // subtrees still to be laid out
        {
            std::vector<uint32_t> bfs; bfs.push_back(0);
            for (size_t h = 0; h < bfs.size() && bfs.size() < QUAD_TOP_NODES; ++h) { std::vector<uint32_t> k; kids(bfs[h], k); for (uint32_t r : k) { if (bfs.size() < QUAD_TOP_NODES) bfs.push_back(r); else roots.push_back(r); } }
            // children of the top's own last records that were never expanded
            std::vector<char> in_top(H.nodes4.size(), 0); for (uint32_t o : bfs) in_top[o] = 1;
            roots.clear();
            for (uint32_t o : bfs) { std::vector<uint32_t> k; kids(o, k); for (uint32_t r : k) if (!in_top[r]) roots.push_back(r); }
            seq = bfs;
        }
        if (node_order == 1) {
            // treelets of three levels, each laid out breadth-first, in the order their roots were met
            for (size_t h = 0; h < roots.size(); ++h) {
                std::vector<uint32_t> level{roots[h]};
                for (int l = 0; l < 3 && !level.empty(); ++l) {
                    std::vector<uint32_t> next;
                    for (uint32_t o : level) { seq.push_back(o); kids(o, next); }
                    level.swap(next);
                }
                for (uint32_t o : level) roots.push_back(o); // the records below the treelet's third level start treelets of their own
            }
        }
        uint32_t next = 0;
        for (uint32_t o : seq) newid[o] = next++;
        for (size_t o = 0; o < H.nodes4.size(); ++o) if (newid[o] == 0xffffffffu) newid[o] = next++;
        std::vector<DNode4> re(H.nodes4.size());
        for (size_t o = 0; o < H.nodes4.size(); ++o) { DNode4 dn = H.nodes4[o]; for (int sl = 0; sl < 4; ++sl) if (dn.ref[sl] != REF_NONE && !(dn.ref[sl] & REF_LEAF)) dn.ref[sl] = newid[dn.ref[sl]]; re[newid[o]] = dn; }
        H.nodes4.swap(re);
    }
    H.tris.resize(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        const DTriShade &T = H.shade[order[k]]; DTri &t = H.tris[k];
        t.p0[0] = T.p0[0]; t.p0[1] = T.p0[1]; t.p0[2] = T.p0[2]; t.p1x = T.p1x; t.p1y = T.p1y; t.p1z = T.p1z;
        t.p2x = T.p2x; t.p2y = T.p2y; t.p2z = T.p2z; t.prim = order[k]; t.flags = T.flags; t.alpha_tex = T.alpha_tex;
    }
    return PTRS_OK;
}

} // namespace pt
