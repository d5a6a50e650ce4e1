// ptrs_hip.hip -- gfx950 (MI355X) kernels and the C ABI of include/ptrs.h.
//
// One wavefront stage = one kernel.  Every queue-driven kernel is PERSISTENT: a launch holds the workgroups that fit the machine at
// once (256 threads = 4 wave64 each), and each WAVE works through queue segments -- one wave per segment -- that it takes from the
// launch's ticket counters until none is left; segment lengths are read from device memory, so a whole pass is enqueued without
// host round trips.  Slots in the segment a wave appends to come from a counter in a scalar register of that wave: no LDS or
// global atomics on the data path, no workgroup barrier behind the staging of the read-only tables.
// The traversal stack lives in LDS, one column of 8-byte (node ref, entry distance) records per lane; on trees deeper than the LDS
// column the excess spills to a per-thread column in global memory.  Small scenes live in LDS entirely, resolved by ray class.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (bit-exact arithmetic, see pt_vec.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pt_render.h"

using namespace pt;

// ---- embedded Sobol' tables (generated from data/sobol_tables.bin by the build) -------------------
static const unsigned char k_sobol_blob[] = {
#include "sobol_tables_data.inc"
};

namespace {

thread_local std::string g_err;

// Process-wide tuning knobs (ptrs_set_option).  The library reads no environment variables: an embedding host sets what
// it needs once; the defaults are the measured best on MI355X.
struct Options {
    int lanes = 0;            // concurrent pipeline lanes (own path state, queues and stream each), 1..MAX_LANES; 0 = by the size of the job (HipBackend::lanes: 4, 2 under 4 M paths, 1 for one pixel)
    int refill = -1;          // idle-lane threshold of the lane-refill extension kernel; 0 = the fused k_extend; -1 = by scene (32 with phase voting and no alpha masks, else 16)
    int refill_connect = -1;  // the same for the connection kernel (it resolves shadow-only NEE records itself; Cornell: fused k_connect 68 ms, refill 48 ms per frame)
    int stack_lds = 8;        // LDS traversal-stack entries per lane for quad-form scenes: 8 (+ tree top cached in LDS) or 16
    int grid_mult = 0;        // queue segments (one wave each) per pass: CUs x 8 x grid_mult; 0 = with several lanes 1 (2 048 on MI355X; 1-4 for scenes whose tree is in HBM, by the paths of a pass: HipBackend::lanes), 8 with one (16 384; Cornell single-lane frame at 4 / 8 / 16: 252 / 216 / 222 ms)
    int grid_pct = 0;         // share of its resident capacity a persistent launch takes: below 100 a kernel leaves wave slots to the kernels of the other pipeline lanes; 0 = 100 (50 with several lanes and a grid_mult given by hand)
    int whole_rounds = 0;     // 1: segments per pass rounded to a whole multiple of the traversal kernels' resident waves (HipBackend::whole_rounds); measured within noise of 0 (exactly CUs x 8 x grid_mult) on Cornell / classroom, 2.5 % slower on colonnade
    int persist = 1;          // queue kernels are launched with the workgroups that fit the machine at once (occupancy x CUs); 0: with 8 per CU, the hardware's maximum (A/B hook)
    int node_form = 0;        // 0 = by size, 2 = quad nodes also for scenes that would fit LDS (test hook)
    int node_order = 0;       // quad form, records behind the LDS-cached top: 0 = the builder's depth-first order, 1 = treelets of three levels (pt_host_scene.h)
    int vote = -1;            // lane-refill traversal kernels: each step runs the phase (node visit / triangle test) most lanes of the wave are in.
                              // 0 off, 1 both kernels, 2 extension kernel only, -1 = by scene: 1 for quad-form scenes (colonnade extend 86 -> 56 ms), 2 for the LDS-resident
                              // pair form, whose cheap steps do not pay for the vote in the connect kernel (Cornell: extend 92.5 -> 89 ms, connect 52 -> 63 ms)
    int fused_epilogue = 1;   // k_extend_rf / k_connect_rf run their segment's epilogue (emission, depth cut, material bucketing) / MIS resolve behind their last ray; 0: the separate k_epilogue / k_resolve
    int fused_resolve = 1;    // (with fused_epilogue) k_connect_rf resolves its MIS records itself
    int shade_lds = 1;        // shade kernels read light records, small scenes' triangle records and the round's Sobol' tables from LDS (0: everything from global memory)
    int env_presample = 1;    // scenes lit by an environment map: the light's samples of a round's vertices are evaluated by k_env_presample ahead of the shade kernels (0: inside them)
    int peer_copy = 1;        // ptrs_render_multi: bands travel device to device (hipMemcpyPeerAsync over xGMI); 0: staged through the host film, the path taken when two devices cannot reach each other (test hook)
    int deal = 0;             // how k_generate deals a pass's 64-path chunks to the queue segments: 0 round-robin (every segment a sample of the whole image: even load), 1 by image region (a segment = a patch of pixels with all its samples; segment ranges follow the ticket counters and those the XCDs: an XCD's L2 serves one region's rays -- measured 4-7 % SLOWER on all three workloads: the regions' paths die at different rates and the segments' loads with them)
    int tail = 1;             // thin late rounds of a pass run in ONE launch (k_tail: every wave takes its segment through all remaining rounds) where the scene has an instantiation; 0: every round is its three launches
    int tail_at = -1;         // the round at which a pass hands over to k_tail: -1 = the first round in which the paths expected alive (this scene's survival profile, learned from its last finished pass) are at most tail_paths per segment; k >= 0: round k
    int tail_paths = 0;       // (tail_at = -1) paths per segment at or below which the tail takes over; 0 = the measured default (TAIL_PATHS_DEFAULT)
    int workspace_pct = 40;   // the render workspace (path state + queues of all lanes) may take this share of the device memory that is free at the call
};
Options g_opt;
std::mutex g_opt_mu;
Options options() { std::lock_guard<std::mutex> lk(g_opt_mu); return g_opt; }
struct OptionDesc { const char *name; int Options::*field; int lo, hi; };
const OptionDesc k_options[] = {
    {"lanes", &Options::lanes, 0, 8}, {"refill", &Options::refill, -1, 64}, {"refill_connect", &Options::refill_connect, -1, 64}, {"stack_lds", &Options::stack_lds, 8, 16},
    {"grid_mult", &Options::grid_mult, 0, 64}, {"persist", &Options::persist, 0, 1}, {"whole_rounds", &Options::whole_rounds, 0, 1}, {"grid_pct", &Options::grid_pct, 0, 100}, {"node_form", &Options::node_form, 0, 2}, {"node_order", &Options::node_order, 0, 1}, {"vote", &Options::vote, -1, 2}, {"shade_lds", &Options::shade_lds, 0, 1}, {"fused_epilogue", &Options::fused_epilogue, 0, 1}, {"fused_resolve", &Options::fused_resolve, 0, 1}, {"workspace_pct", &Options::workspace_pct, 1, 90}, {"peer_copy", &Options::peer_copy, 0, 1}, {"env_presample", &Options::env_presample, 0, 1}, {"deal", &Options::deal, 0, 1}, {"tail", &Options::tail, 0, 1}, {"tail_at", &Options::tail_at, -1, 64}, {"tail_paths", &Options::tail_paths, 0, 1 << 20},
};

#define HIPCHK(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            g_err = std::string(#expr) + ": " + hipGetErrorString(e_);                                            \
            return PTRS_ERR_DEVICE;                                                                               \
        }                                                                                                         \
    } while (0)

constexpr int BLOCK = 256, WAVES = BLOCK / 64;

// Segmented queues, one WAVE per segment.  Every queue of a pass is split into G segments; the wave that consumes segment s of
// one stage appends only to segment s of the next, so a segment has one writer at a time and its slots come from a counter the
// wave keeps in a scalar register (one s_bcnt1 per push): no atomics on the data path at all, neither global (the first version:
// ~16 M returning atomics per frame on one address) nor LDS (round 2: one ds_add per wave and push, plus a workgroup barrier
// per segment).  The queue kernels are PERSISTENT: a launch holds only as many workgroups as fit the machine at once
// (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs), and each of their waves takes segment numbers from a per-launch ticket
// counter until the G segments are gone (one returning global atomic per wave and segment, asked for one segment ahead).  With a
// workgroup per segment and G = 2048 workgroups on 5-7 resident per CU, the second scheduling round of every launch ran a third
// full (time-average occupancy 8 / ceil(8 / c) waves per SIMD); now every wave slot works until the tickets run out, and the
// tail is one wave-segment (G / resident waves is 1.3-2.7 at the default grid_mult).
__device__ inline uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline uint32_t lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); } // set bits of m below this lane
__device__ inline uint32_t wave_push(uint32_t &count, bool pred) { // slot for every lane with pred; count is wave-uniform
    const unsigned long long m = __ballot(pred);
    const uint32_t slot = count + lanes_below(m);
    count += (uint32_t)__popcll(m);
    return slot;
}
// The segment a wave works on next, or a number >= G when the launch has none left.  One counter for the whole launch would take one
// returning atomic per segment on ONE address, ~18 ns apiece on MI355X: 0.3 ms for G = 16384, which is what a kernel of the thin late
// rounds of a pass then costs however little it has to do (measured: 11 of Cornell's 16 rounds, 30 ms of a 200 ms frame).  So the
// launch has TK_SUB counters 128 bytes apart, each handing out a contiguous range of ceil(G / TK_SUB) segments; a wave reads all of
// them with one load (lane k reads counter k), takes a ticket from the first open one at or after its home counter, and moves on as
// ranges run out.  The atomics of a launch spread over TK_SUB addresses, and a wave sees the launch exhausted with a single load.
enum : uint32_t { TK_SUB = 64, TK_SUB_STRIDE = 32, TK_LAUNCH_WORDS = TK_SUB * TK_SUB_STRIDE };
__device__ inline uint32_t seg_next(uint32_t *ticket, uint32_t G) {
    const uint32_t per = (G + TK_SUB - 1u) / TK_SUB; // every counter hands out `per` numbers; those at or beyond G (the last counters' surplus) are skipped
    const uint32_t home = (blockIdx.x * WAVES + (threadIdx.x >> 6)) & (TK_SUB - 1u);
    for (;;) {
        // lane k reads counter k (a device-coherent load: the counters only grow, a stale value costs one atomic that comes back empty); which
        // counters are still open is a lane mask.  In one asm statement with one temporary: written in C++ the scan's per-lane values are
        // hoisted out of the callers' segment loops and cost every kernel ~10 vector registers.
        uint32_t tmp; unsigned long long open;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 7, %0\n\tglobal_load_dword %0, %0, %2 sc1\n\ts_waitcnt vmcnt(0)\n\tv_cmp_gt_u32 %1, %3, %0"
                     : "=&v"(tmp), "=s"(open) : "s"(ticket), "s"(per) : "memory");
        static_assert(TK_SUB_STRIDE * 4u == 128u, "the asm above shifts the lane id by 7");
        if (open == 0ull) return 0xffffffffu;
        const unsigned long long rot = home ? ((open >> home) | (open << (64u - home))) : open; // the home counter first
        const uint32_t c = (home + (uint32_t)__ffsll((long long)rot) - 1u) & (TK_SUB - 1u);
        uint32_t t = 0;
        if (__lane_id() == 0) t = atomicAdd(ticket + c * TK_SUB_STRIDE, 1u);
        t = rfl(t);
        const uint32_t s = c * per + t;
        if (t < per && s < G) return s;
        // the counter ran out between the load and the atomic, or the number is one of its surplus: look again
    }
}
// counts[(row * Q_STRIDE + q) * G + s]
__device__ inline uint32_t *seg_count(const DQueues &Q, uint32_t row, int q, uint32_t G, uint32_t s) { return Q.counts + ((size_t)row * Q_STRIDE + (size_t)q) * G + s; }
enum { TK_EXTEND = 0, TK_CONNECT = 1, TK_SHADE0 = 2, TK_EPILOGUE = 9, TK_RESOLVE = 10, TK_PRESAMPLE = 11, TK_TAIL = 12 }; // tickets[(row * Q_STRIDE + TK_*) * TK_LAUNCH_WORDS]: the counters of one launch of a pass
// A round nobody reaches (every path has ended: rounds are enqueued without asking, and for scenes with null-BSDF skips a few more
// than max_depth + 1) costs its launches only: each kernel looks at the flag and leaves.
// (Q.alive[row]: the round's "some path is still alive" flag, set by the shade kernels of the round before)
__device__ inline bool round_is_dead(const DQueues &Q, uint32_t it) { return it > 0u && rfl(Q.alive[it]) == 0u; }

// Geometry sources of the traversal kernels.  LDS pointers keep their address space in the type: with two generic pointers the
// compiler folds both paths into one generic-address (flat) load, which is slower than either.
// Records sit in LDS at an ODD number of 16-byte vectors apart (7 per LDS-form node, 9 per quad node): a ds_read_b128 is served in
// four groups of 16 lanes, 256 bytes = 16 vector slots per LDS cycle, and lanes of a group that read different records conflict
// when their slots coincide.  At the records' natural strides (4 and 8 vectors) all pair nodes share 4 slots per vector and all
// quad nodes 2 (measured: 2.4-3.7 conflict cycles per LDS instruction); at an odd stride 16 consecutive records cover all 16 slots
// (0.5-0.6 measured).
typedef __attribute__((address_space(3))) const v4 lds_v4;
__device__ inline v4 lds_ld(lds_v4 *q) { v4 r; r.x = q->x; r.y = q->y; r.z = q->z; r.w = q->w; return r; }
enum : uint32_t { TOP_LDS_STRIDE = 9 };
// Quad nodes with the top of the tree (the first `ktop` records, breadth-first) in LDS and the rest in global memory.
struct GeomTop {
    lds_v4 *top; const v4 *nodesv; const DTri *tris; uint32_t ktop;
    __device__ inline void node8(uint32_t i, v4 *o) const {
        if (i < ktop) { lds_v4 *q = top + TOP_LDS_STRIDE * i; for (int k = 0; k < 8; ++k) o[k] = lds_ld(q + k); }
        else { const v4 *q = nodesv + 8u * i; for (int k = 0; k < 8; ++k) o[k] = q[k]; }
    }
    // (see GeomGlobal::quad_load; the top of the tree out of LDS, records 9 vectors apart)
    __device__ inline void quad_load(uint32_t i, uint32_t px, uint32_t py, uint32_t pz, v4 &xn, v4 &xf, v4 &yn, v4 &yf, v4 &zn, v4 &zf, v4 &refs, v4 &meta) const {
        if (i < ktop) {
            typedef __attribute__((address_space(3))) const char lds_c;
            lds_c *b = (lds_c *)top + i * (TOP_LDS_STRIDE * 16u);
            xn = lds_ld((lds_v4 *)(b + px)); xf = lds_ld((lds_v4 *)(b + (px ^ 16u)));
            yn = lds_ld((lds_v4 *)(b + py)); yf = lds_ld((lds_v4 *)(b + (py ^ 16u)));
            zn = lds_ld((lds_v4 *)(b + pz)); zf = lds_ld((lds_v4 *)(b + (pz ^ 16u)));
            refs = lds_ld((lds_v4 *)(b + 96u)); meta = lds_ld((lds_v4 *)(b + 112u));
        } else {
            const char *b = reinterpret_cast<const char *>(nodesv);
            const uint32_t off = i * 128u;
            xn = *reinterpret_cast<const v4 *>(b + (size_t)(off + px)); xf = *reinterpret_cast<const v4 *>(b + (size_t)((off + px) ^ 16u));
            yn = *reinterpret_cast<const v4 *>(b + (size_t)(off + py)); yf = *reinterpret_cast<const v4 *>(b + (size_t)((off + py) ^ 16u));
            zn = *reinterpret_cast<const v4 *>(b + (size_t)(off + pz)); zf = *reinterpret_cast<const v4 *>(b + (size_t)((off + pz) ^ 16u));
            refs = *reinterpret_cast<const v4 *>(b + (size_t)(off + 96u)); meta = *reinterpret_cast<const v4 *>(b + (size_t)(off + 112u));
        }
    }
    __device__ inline void node(uint32_t, v4 &, v4 &, v4 &, v4 &) const {}
    __device__ inline void tri(uint32_t k, v4 &a, v4 &b, v4 &c) const { const v4 *q = reinterpret_cast<const v4 *>(tris + k); a = q[0]; b = q[1]; c = q[2]; }
};

// Traversal stack: column `threadIdx.x` of a [D][BLOCK] LDS array of (ref, entry distance) records.
// OVF: entries beyond D go to this thread's column of a global array (StackSpill), so deep trees keep
// the LDS footprint -- and with it the occupancy -- of a 16-entry stack.
struct StackSpill { unsigned long long *p; uint32_t stride; };
template <int D, bool OVF>
struct LdsStack { // records are packed into one 64-bit word (ref | entry distance << 32): one ds_write_b64 / ds_read_b64 each
    unsigned long long *col, *ovf; uint32_t ovf_stride;
    int n;
    __device__ inline void init(unsigned long long *lds, const StackSpill &sp) { col = lds + threadIdx.x; ovf = OVF ? sp.p + (size_t)blockIdx.x * BLOCK + threadIdx.x : nullptr; ovf_stride = sp.stride; n = 0; }
    __device__ inline void push(uint32_t v, float t) {
        const unsigned long long e = (unsigned long long)v | ((unsigned long long)f2u(t) << 32);
        if (!OVF || n < D) col[n * BLOCK] = e; else ovf[(size_t)(n - D) * ovf_stride] = e;
        ++n;
    }
    static constexpr bool select_push = !OVF; // (a stack without overflow can push in select form: the store goes out for every lane, the count moves for those that push)
    __device__ inline void push_if(bool yes, uint32_t v, float t) { col[n * BLOCK] = (unsigned long long)v | ((unsigned long long)f2u(t) << 32); n += yes ? 1 : 0; }
    __device__ inline void pop(uint32_t &v, float &t) {
        --n;
        unsigned long long e = col[(OVF && n >= D ? D - 1 : n) * BLOCK]; // always an LDS read (no generic-address select); the spill is the rare path
        if (OVF && n >= D) e = *(volatile unsigned long long *)(ovf + (size_t)(n - D) * ovf_stride); // volatile: keeps the compiler from folding both reads into one generic-address load
        v = (uint32_t)e; t = u2f((uint32_t)(e >> 32));
    }
    __device__ inline bool empty() const { return n == 0; }
    __device__ inline void clear() { n = 0; }
};

// ---- kernels ----------------------------------------------------------------------------------------
// Wave s generates the paths of segment s.  The pass's path slots -- sample-major: slot = sample x pixels + pixel -- are cut into chunks
// of 64 consecutive slots (coalesced state access per wave: 1 KB per state array and chunk), and the chunks are dealt to the G segments
//   deal = 0: round-robin (chunk c to segment c mod G): every segment holds paths of the whole image;
//   deal = 1: BY IMAGE REGION: the chunks form a matrix (rows: samples, columns: the ~pixels / 64 chunks of one sample's pixels) that
//             is walked column by column, and segment s takes the s-th run of ceil(chunks / G) cells: a segment holds a few columns --
//             a patch of neighbouring pixels -- with ALL their samples, and neighbouring segments neighbouring patches.  Segments keep
//             their identity through every stage of every round (the wave that consumes segment s appends only to segment s), the
//             ticket counters hand out contiguous segment ranges, and a workgroup's home counter follows blockIdx mod 16, i.e. its XCD
//             (workgroups are placed round-robin over the 8 XCDs): an XCD's private L2 then serves the rays of an eighth of the image
//             -- primary and shadow rays of one region walk the same part of the tree -- instead of a sample of all of it.  When a
//             range runs dry its waves move on to the next counter's (the neighbouring region).  Placement is a matter of speed only.
// Either way a segment's entries are compact (positions e0 .. e0 + n) and hold at most seg_cap = ceil(chunks / G) x 64 paths.
__global__ __launch_bounds__(BLOCK) void k_generate(DParams R, DSampler S, DCamera C, DPaths P, DQueues Q, uint32_t seg_cap, uint32_t G, uint32_t deal) {
    const uint32_t lane = threadIdx.x & 63u, s = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (s >= G) return;
    const uint32_t chunks = (R.n_paths + 63u) / 64u;
    const uint32_t e0 = s * seg_cap; // the segment's first position in the queue-ordered arrays
    uint32_t n = 0;
    if (!deal) {
        for (uint32_t c = s; c < chunks; c += G) {
            const uint32_t pid = c * 64u + lane;
            if (pid < R.n_paths) {
                const uint32_t e = e0 + (c / G) * 64u + lane;
                generate_item(R, S, C, P, pid, e);
                pslot(Q.ext[0], e) = pid;
            }
            n += (c * 64u + 64u <= R.n_paths) ? 64u : (R.n_paths - c * 64u);
        }
    } else {
        const uint32_t npix = R.pixel_mode ? 1u : (uint32_t)(R.row1 - R.row0) * (uint32_t)R.NX;
        uint32_t cpp = (npix + 63u) / 64u; // columns: the chunks of one sample's pixels (when pixels % 64 != 0 a row drifts by a fraction of a chunk per sample: nothing to a region)
        if (cpp > chunks) cpp = chunks;
        if (cpp < 1u) cpp = 1u;
        const uint32_t nsr = (chunks + cpp - 1u) / cpp;          // rows
        const uint32_t r_last = chunks - (nsr - 1u) * cpp;       // the columns that have a cell in the last row
        const uint32_t full = r_last * nsr;                      // cells of those columns in column-major order; the columns behind them have nsr - 1 cells
        const uint32_t cps = seg_cap / 64u;
        const uint32_t c1 = (s + 1u) * cps < chunks ? (s + 1u) * cps : chunks;
        for (uint32_t cc = s * cps; cc < c1; ++cc) {
            uint32_t pc, k;
            if (cc < full) { pc = cc / nsr; k = cc - pc * nsr; }
            else { const uint32_t d = cc - full, q = d / (nsr - 1u); pc = r_last + q; k = d - q * (nsr - 1u); } // (full < chunks only when nsr >= 2)
            const uint32_t c = k * cpp + pc, pid = c * 64u + lane;
            if (pid < R.n_paths) {
                const uint32_t e = e0 + n + lane; // (only the pass's last chunk is partial, and its paths are its first lanes: positions stay compact)
                generate_item(R, S, C, P, pid, e);
                pslot(Q.ext[0], e) = pid;
            }
            n += (c * 64u + 64u <= R.n_paths) ? 64u : (R.n_paths - c * 64u);
        }
    }
    if (lane == 0) *seg_count(Q, 0, Q_EXT, G, s) = n;
}

// Ray source: (ro, rd) indexed by path slot, ro.w = t_max.  ANY: write occl[pid]; else write hits[pid].
template <bool ANY, int DEPTH, bool OVF>
__global__ __launch_bounds__(BLOCK) void k_trace(DScene sc, StackSpill spill, const uint32_t *__restrict__ queue, const uint32_t *__restrict__ count,
                                                 const v4 *__restrict__ ro, const v4 *__restrict__ rd, u4 *__restrict__ hits,
                                                 uint32_t *__restrict__ occl, float *__restrict__ tout, unsigned long long *stats, uint32_t counters_on) {
    __shared__ unsigned long long lds_stack[DEPTH * BLOCK];
    const uint32_t n = *count;
    const uint32_t stride = gridDim.x * BLOCK;
    uint32_t nn = 0, nt = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t pid = queue ? queue[i] : i;
        const v4 o = ro[pid], d = rd[pid];
        LdsStack<DEPTH, OVF> stk; stk.init(lds_stack, spill);
        HitRec h;
        const bool hit = bvh_trace<ANY>(sc, xyz(o), xyz(d), o.w, stk, h, nn, nt);
        if (ANY) occl[pid] = hit ? 1u : 0u;
        else { u4 r; r.x = (uint32_t)h.prim; r.y = f2u(h.b0); r.z = f2u(h.b1); r.w = f2u(h.b2); hits[pid] = r; if (tout) tout[pid] = h.t; }
    }
    if (counters_on) { atomicAdd(&stats[CNT_NODES], (unsigned long long)nn); atomicAdd(&stats[CNT_TRIS], (unsigned long long)nt); }
}

// One step of every lane's ray.  vote = 0: the while-while loop (descend to a leaf with all lanes that are still at inner
// nodes, then that leaf's triangles): a wave waits for its slowest descent and its fattest leaf.  vote = 1: the wave looks at
// what its lanes need next -- a node visit or a triangle test -- and runs the one more lanes are waiting for, ONE visit or ONE
// triangle; the others sit the step out.  No lane waits for a straggler, only for its phase to get the majority (which it
// does: every descending lane reaches a leaf), so at least half of the lanes holding rays work in every step.  Each ray still
// makes exactly the visits and tests of the while-while loop in the same order: results are identical.
// (Measured, single lane, VALU lanes active per instruction / ms per frame: see DESIGN.md section 4.)
// Diagnostic builds (-DPTRS_STEP_COUNTERS, tools/ablate.sh + tools/occupancy.py) count the wave-level node / triangle steps
// and the lanes that took part in them (64 per step, added by the wave's first active lane); the product carries none of it.
struct StepCount { uint32_t node_steps = 0, node_visits = 0, tri_steps = 0; };
#ifdef PTRS_STEP_COUNTERS
__device__ inline uint32_t first_lane_64() { return (int)__lane_id() == __ffsll((long long)__ballot(true)) - 1 ? 64u : 0u; }
#define PT_COUNT_NODE(c) { (c).node_steps += first_lane_64(); ++(c).node_visits; }
#define PT_COUNT_TRI(c, k) { (c).tri_steps += first_lane_64() * (k); }
#else
#define PT_COUNT_NODE(c)
#define PT_COUNT_TRI(c, k)
#endif
// ---- LDS form: a small scene resolved by ray class --------------------------------------------------------------------------
// On gfx950 only fp32 add / sub / mul / fma and the plain bit operations issue at the full rate (one wave64 instruction per
// ~2.2 cycles and SIMD, two waves side by side); comparisons, v_cndmask, min / max, shifts, integer mads, fp64 and packed fp32 go
// through one port at one per ~4.2 cycles (tools/valu_ceiling.hip, profiles/r03_valu_ceiling.json).  The pair-node visit of round 2
// spent 45 of its 95 vector instructions in that class: twelve selects that pick each axis' near and far plane by the sign of the
// ray direction, fourteen comparisons, eight min / max, ten more for the visiting order; the triangle test 18 selects for the
// permutation that makes |d| largest in z.  A scene that lives in LDS can hold the result of those selects instead:
//   * node i = 7 vectors: per axis the record (child 0 near, child 0 far, child 1 near, child 1 far) twice -- for rays going up
//     and for rays going down that axis -- and (ref0, ref1, 1 << split axis, -); a lane reads the three records of ITS signs
//     (the byte offsets are per-ray constants) and never selects a plane.  A missing second child is the box (+inf, -inf).
//   * references are LDS byte addresses (interior) or REF_LEAF | (count - 1) << 27 | byte offset of the leaf's first triangle
//     record: no index arithmetic in the loop.
//   * the triangles' 48-byte records three times, vertex components rotated so that (kx, ky, kz) come first, second, third for
//     kz = 0, 1, 2; the ray keeps its origin permuted the same way and the address of its copy.
// Cornell: 29 nodes x 112 B + 3 x 36 x 48 B = 8.4 KB.  Same arithmetic on the same values in the same order as pair_visit /
// leaf_step: same bits (test_traversal_kernel_variants..., test_rays_inside_box_planes).
enum : uint32_t { LN_V4 = 7, LN_BYTES = LN_V4 * 16u, LT_BYTES = 48u };
struct LdsGeom { uint32_t root, tri0, tri_copy; }; // LDS byte addresses: node 0, triangle copy 0; bytes per triangle copy
typedef __attribute__((address_space(3))) const u4 lds_u4;
__device__ inline v4 lds_at(uint32_t addr) { return lds_ld((lds_v4 *)(uintptr_t)addr); }
__device__ inline uint32_t lds_form_ref(uint32_t ref, uint32_t base) {
    if (ref == REF_NONE) return REF_NONE;
    if (ref & REF_LEAF) return (ref & ~(uint32_t)REF_FIRST_MASK) | ((ref & REF_FIRST_MASK) * LT_BYTES);
    return base + ref * LN_BYTES;
}
// GEOM = capacity in vectors; the host guarantees LN_V4 * n_nodes2 + 9 * n_prims <= GEOM.  The caller's barrier makes the copy visible.
__device__ inline LdsGeom stage_lds_form(const DScene &sc, v4 *lds) {
    const uint32_t base = (uint32_t)(uintptr_t)(lds_v4 *)lds;
    const uint32_t nn = sc.n_nodes2, nt = sc.n_prims;
    for (uint32_t w = threadIdx.x; w < nn * LN_V4; w += BLOCK) {
        const uint32_t i = w / LN_V4, v = w - i * LN_V4;
        const float *f = reinterpret_cast<const float *>(sc.nodes2 + i); // c0min[3] c0max[3] c1min[3] c1max[3] ref0 ref1 axis -
        const uint32_t ref0 = f2u(f[12]), ref1 = f2u(f[13]), axis = f2u(f[14]);
        v4 r;
        if (v < 6u) {
            const uint32_t a = v >> 1;
            const bool down = (v & 1u) != 0;
            const float lo0 = f[a], hi0 = f[3u + a];
            const float lo1 = ref1 != REF_NONE ? f[6u + a] : PT_INF, hi1 = ref1 != REF_NONE ? f[9u + a] : -PT_INF;
            r.x = down ? hi0 : lo0; r.y = down ? lo0 : hi0; r.z = down ? hi1 : lo1; r.w = down ? lo1 : hi1;
        } else { r.x = u2f(lds_form_ref(ref0, base)); r.y = u2f(lds_form_ref(ref1, base)); r.z = u2f(axis < 3u ? 1u << axis : 0u); r.w = 0.0f; }
        lds[w] = r;
    }
    v4 *lt = lds + nn * LN_V4;
    for (uint32_t w = threadIdx.x; w < 9u * nt; w += BLOCK) {
        const uint32_t kz = w / (3u * nt), rem = w - kz * 3u * nt, k = rem / 3u, v = rem - k * 3u;
        const uint32_t kx = kz == 2u ? 0u : kz + 1u, ky = kx == 2u ? 0u : kx + 1u;
        const float *f = reinterpret_cast<const float *>(sc.tris + k); // p0[3] p1[3] p2[3] prim flags alpha_tex
        float o[4];
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t d = 4u * v + j; // dword of the record
            if (d < 9u) { const uint32_t vert = d / 3u, c = d - vert * 3u; o[j] = f[vert * 3u + (c == 0u ? kx : (c == 1u ? ky : kz))]; }
            else o[j] = f[d];
        }
        v4 r; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.w = o[3];
        lt[w] = r;
    }
    LdsGeom G; G.root = base; G.tri0 = base + nn * LN_BYTES; G.tri_copy = nt * LT_BYTES;
    return G;
}
// Per-lane state of an LDS-form ray, plain scalars (see RF_DECL): origin, 1 / d, the origin permuted to (kx, ky, kz), the shear,
// the byte offsets of the ray's three plane records inside a node, its sign bits, the address of its triangle copy.
#define LF_DECL f3 l_o = mk3(0, 0, 0), l_inv = mk3(1, 1, 1), l_op = mk3(0, 0, 0); float l_sx = 0.0f, l_sy = 0.0f, l_sz = 1.0f; uint32_t l_ox = 0, l_oy = 32, l_oz = 64, l_neg = 0, l_tri = 0;
#define LF_START(LG, O, D, TMAX) { l_o = (O); const f3 d_ = (D); r_tmax = (TMAX); l_inv = mk3(1.0f / d_.x, 1.0f / d_.y, 1.0f / d_.z); \
                const bool nx_ = l_inv.x < 0.0f, ny_ = l_inv.y < 0.0f, nz_ = l_inv.z < 0.0f; l_neg = (nx_ ? 1u : 0u) | (ny_ ? 2u : 0u) | (nz_ ? 4u : 0u); \
                l_ox = nx_ ? 16u : 0u; l_oy = ny_ ? 48u : 32u; l_oz = nz_ ? 80u : 64u; \
                const RayShear sh_ = ray_shear_inv(d_, l_inv); l_sx = sh_.sx; l_sy = sh_.sy; l_sz = sh_.sz; \
                const int kx_ = sh_.kz == 2 ? 0 : sh_.kz + 1, ky_ = kx_ == 2 ? 0 : kx_ + 1; l_op = mk3(comp(l_o, kx_), comp(l_o, ky_), comp(l_o, sh_.kz)); \
                l_tri = (LG).tri0 + (uint32_t)sh_.kz * (LG).tri_copy; \
                r_h.prim = -1; r_h.t = r_tmax; r_h.b0 = r_h.b1 = r_h.b2 = 0.0f; r_h.flags = 0; r_hit = false; r_cur = (LG).root; }

// pair_visit on the LDS form: both children's slab tests from the ray's three plane records, the far child stacked with its entry
// distance when both are hit (pop-time re-test, accelerator.rs:372), the visiting order from the split axis' sign bit.
template <class Stack>
__device__ inline void lf_node_visit(uint32_t &cur, f3 o, f3 inv, uint32_t ox, uint32_t oy, uint32_t oz, uint32_t negbits, float t_max, Stack &stack, uint32_t &n_nodes) {
    const v4 X = lds_at(cur + ox), Y = lds_at(cur + oy), Z = lds_at(cur + oz), Rf = lds_at(cur + 96u);
    const float k = 1.0f + 2.0f * gamma_err(3);
    n_nodes += 2;
    float t0, t1;
    bool h0 = slab_finish((X.x - o.x) * inv.x, (Y.x - o.y) * inv.y, (Z.x - o.z) * inv.z, ((X.y - o.x) * inv.x) * k, ((Y.y - o.y) * inv.y) * k, ((Z.y - o.z) * inv.z) * k, t0);
    bool h1 = slab_finish((X.z - o.x) * inv.x, (Y.z - o.y) * inv.y, (Z.z - o.z) * inv.z, ((X.w - o.x) * inv.x) * k, ((Y.w - o.y) * inv.y) * k, ((Z.w - o.z) * inv.z) * k, t1);
    h0 = h0 & (t0 < t_max); h1 = h1 & (t1 < t_max);
    const uint32_t ref0 = f2u(Rf.x), ref1 = f2u(Rf.y);
    const bool second_first = (negbits & f2u(Rf.z)) != 0u;
    // (h0 & h1, h0 | h1 on the lane masks themselves -- s_and_b64 / s_or_b64 -- instead of the 0 / 1 integers in vector registers the
    // compiler makes of two booleans that feed several branches)
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
    if (Stack::select_push) { // no divergent region around the push and the one-child case: selects, one unconditional LDS store, and the pop as the only branch
        const bool both = __builtin_amdgcn_inverse_ballot_w64(m0 & m1);
        stack.push_if(both, second_first ? ref0 : ref1, second_first ? t0 : t1);
        cur = both ? (second_first ? ref1 : ref0) : (h0 ? ref0 : ref1);
        if (!__builtin_amdgcn_inverse_ballot_w64(m0 | m1)) cur = pop_next_ref<false>(stack, t_max);
        return;
    }
    if (__builtin_amdgcn_inverse_ballot_w64(m0 & m1)) { stack.push(second_first ? ref0 : ref1, second_first ? t0 : t1); cur = second_first ? ref1 : ref0; }
    else if (__builtin_amdgcn_inverse_ballot_w64(m0 | m1)) cur = h0 ? ref0 : ref1;
    else cur = pop_next_ref<false>(stack, t_max);
}
// leaf_step on the LDS form: ONE triangle of the leaf, from the ray's permuted copy.
template <bool ALPHA>
__device__ inline bool lf_leaf_step(const DScene &sc, uint32_t &leaf, f3 op, float sx, float sy, float sz, uint32_t tri_base, float &t_max, HitRec &out, bool &hit, uint32_t &n_tris, bool any_rt) {
    const uint32_t first = leaf & REF_FIRST_MASK, rest = (leaf >> REF_COUNT_SHIFT) & 15u; // byte offset of this triangle's record; triangles after this one
    leaf = rest ? leaf - (1u << REF_COUNT_SHIFT) + LT_BYTES : REF_NONE;
    const v4 ta = lds_at(tri_base + first), tb = lds_at(tri_base + first + 16u), tc = lds_at(tri_base + first + 32u);
    const uint32_t prim = f2u(tc.y), flags = f2u(tc.z);
    ++n_tris;
    TriHit h;
    const bool ok = tri_test_perm_sel(mk3(ta.x - op.x, ta.y - op.y, ta.z - op.z), mk3(ta.w - op.x, tb.x - op.y, tb.y - op.z), mk3(tb.z - op.x, tb.w - op.y, tc.x - op.z), sx, sy, sz, t_max, h) && !(flags & TRI_DEGENERATE);
    if (ALPHA) { // (alpha-masked meshes: the texture lookup stays behind a branch)
        if (ok) {
            if ((flags & TRI_HAS_ALPHA) && alpha_rejects(sc, prim, (int32_t)f2u(tc.w), h)) return false;
            if (any_rt) { out.prim = 0; hit = true; return true; }
            hit = true; t_max = h.t;
            out.prim = (int32_t)prim; out.t = h.t; out.b0 = h.b0; out.b1 = h.b1; out.b2 = h.b2; out.flags = flags;
        }
        return false;
    }
    // the hit record in select form as well (see tri_test_perm_sel; Cornell 138.9 -> 138.4 ms, ABAB x 3 + 2): a closest-hit query keeps the nearer hit, an any-hit query
    // is done at its first
    hit = hit | ok;
    t_max = ok ? h.t : t_max;
    out.prim = ok ? (any_rt ? 0 : (int32_t)prim) : out.prim; out.t = ok ? h.t : out.t; out.b0 = ok ? h.b0 : out.b0; out.b1 = ok ? h.b1 : out.b1; out.b2 = ok ? h.b2 : out.b2; out.flags = ok ? flags : out.flags;
    return ok & any_rt;
}
// rf_step for the LDS form (vote: the phase most lanes are in, one visit or one triangle; else descend to a leaf, then its triangles)
template <bool VOTE, bool ALPHA, class Stack>
__device__ inline void lf_step(const DScene &sc, uint32_t &r_cur, f3 l_o, f3 l_inv, f3 l_op, float l_sx, float l_sy, float l_sz, uint32_t l_ox, uint32_t l_oy, uint32_t l_oz, uint32_t l_neg, uint32_t l_tri,
                               float &r_tmax, HitRec &r_h, bool &r_hit, Stack &stk, uint32_t &nn, uint32_t &nt, bool any_rt, StepCount &sc_n) {
    if (VOTE) {
        const bool at_node = (int32_t)r_cur >= 0, at_leaf = (int32_t)r_cur < -1; // leaf references have bit 31 set; lanes without a ray hold REF_NONE = -1
        const uint32_t n_node = rfl((uint32_t)__popcll(__ballot(at_node))), n_leaf = rfl((uint32_t)__popcll(__ballot(at_leaf))); // (scalar registers: the comparison is an s_cmp)
        if (n_node >= n_leaf) {
            if (at_node) { PT_COUNT_NODE(sc_n) lf_node_visit(r_cur, l_o, l_inv, l_ox, l_oy, l_oz, l_neg, r_tmax, stk, nn); }
        } else if (at_leaf) {
            PT_COUNT_TRI(sc_n, 1u)
            const bool done = lf_leaf_step<ALPHA>(sc, r_cur, l_op, l_sx, l_sy, l_sz, l_tri, r_tmax, r_h, r_hit, nt, any_rt);
            if (done) r_cur = REF_NONE; else if (r_cur == REF_NONE) r_cur = pop_next_ref<false>(stk, r_tmax);
        }
        return;
    }
    while (r_cur != REF_NONE && !(r_cur & REF_LEAF)) { PT_COUNT_NODE(sc_n) lf_node_visit(r_cur, l_o, l_inv, l_ox, l_oy, l_oz, l_neg, r_tmax, stk, nn); }
    if (r_cur != REF_NONE) {
        bool done = false;
        while (!done && r_cur != REF_NONE) { PT_COUNT_TRI(sc_n, 1u) done = lf_leaf_step<ALPHA>(sc, r_cur, l_op, l_sx, l_sy, l_sz, l_tri, r_tmax, r_h, r_hit, nt, any_rt); }
        r_cur = done ? REF_NONE : pop_next_ref<false>(stk, r_tmax);
    }
}

__device__ inline GeomTop stage_top(const DScene &sc, v4 *lds, bool top) {
    const GeomGlobal GG0 = geom_global(sc);
    GeomTop GG; GG.top = (lds_v4 *)lds; GG.nodesv = GG0.nodesv; GG.tris = GG0.tris; GG.ktop = 0;
    if (top) { GG.ktop = sc.n_nodes4 < QUAD_TOP_NODES ? sc.n_nodes4 : (uint32_t)QUAD_TOP_NODES; for (uint32_t i = threadIdx.x; i < 8u * GG.ktop; i += BLOCK) lds[TOP_LDS_STRIDE * (i >> 3) + (i & 7u)] = GG0.nodesv[i]; }
    return GG;
}

// ---- lane-refill traversal kernels ------------------------------------------------------------------------------------
// A wave that traced 64 rays in lock step would run as long as its longest ray while finished lanes idle: on a 262 k-triangle
// scene only 21 % of the VALU lanes were active (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU).  Here a lane that finishes its ray
// stores the hit and, once at least `thresh` lanes of the wave are idle, takes the next ray of the wave's queue segment (the
// cursor is a scalar register of the wave).  The per-path epilogue (emission, environment, material bucketing) would run with
// a handful of lanes each time, so it runs behind the segment's last ray with full waves (epilogue_wave).
// Per-lane ray state of the refill kernels: plain scalars on purpose (a struct with the sign array in it made hipcc
// produce a 20 % slower loop).
#define RF_DECL f3 r_o = mk3(0, 0, 0), r_inv = mk3(1, 1, 1); bool r_neg[3] = {false, false, false}; RayShear r_shear; r_shear.kz = 2; r_shear.sx = r_shear.sy = 0.0f; r_shear.sz = 1.0f; \
                float r_tmax = 0.0f; bool r_hit = false; HitRec r_h; r_h.prim = -1; r_h.t = 0.0f; r_h.b0 = r_h.b1 = r_h.b2 = 0.0f; r_h.flags = 0; uint32_t r_cur = REF_NONE, r_nb3 = 0, r_px = 0, r_py = 32, r_pz = 64;
#define RF_START(O, D, TMAX) { r_o = (O); const f3 d_ = (D); r_tmax = (TMAX); r_inv = mk3(1.0f / d_.x, 1.0f / d_.y, 1.0f / d_.z); \
                r_neg[0] = r_inv.x < 0.0f; r_neg[1] = r_inv.y < 0.0f; r_neg[2] = r_inv.z < 0.0f; r_nb3 = neg_bits3(r_neg); r_px = quad_near_x(r_neg); r_py = quad_near_y(r_neg); r_pz = quad_near_z(r_neg); r_shear = ray_shear_inv(d_, r_inv); \
                r_h.prim = -1; r_h.t = r_tmax; r_h.b0 = r_h.b1 = r_h.b2 = 0.0f; r_h.flags = 0; r_hit = false; r_cur = 0; }

template <bool VOTE, bool QUAD, bool ALPHA, class Stack, class Geom>
__device__ inline void rf_step(const Geom &G, const DScene &sc, uint32_t &r_cur, f3 r_o, f3 r_inv, const bool r_neg[3], uint32_t r_nb3, uint32_t r_px, uint32_t r_py, uint32_t r_pz, const RayShear &r_shear, float &r_tmax, HitRec &r_h, bool &r_hit,
                               Stack &stk, uint32_t &nn, uint32_t &nt, bool any_rt, StepCount &sc_n) {
    if (VOTE) {
        const bool at_node = (int32_t)r_cur >= 0, at_leaf = (int32_t)r_cur < -1; // leaf references have bit 31 set; lanes without a ray hold REF_NONE = -1
        const uint32_t n_node = rfl((uint32_t)__popcll(__ballot(at_node))), n_leaf = rfl((uint32_t)__popcll(__ballot(at_leaf))); // (scalar registers: the comparison is an s_cmp)
        if (n_node >= n_leaf) {
            if (at_node) { PT_COUNT_NODE(sc_n) node_visit<QUAD, false>(G, r_cur, r_o, r_inv, r_neg, r_tmax, stk, nn, r_nb3, r_px, r_py, r_pz); }
        } else if (at_leaf) {
            PT_COUNT_TRI(sc_n, 1u)
            const bool done = leaf_step<ALPHA>(G, sc, r_cur, r_o, r_shear, r_tmax, r_h, r_hit, nt, any_rt);
            if (done) r_cur = REF_NONE; else if (r_cur == REF_NONE) r_cur = pop_next_ref<false>(stk, r_tmax);
        }
        return;
    }
    while (r_cur != REF_NONE && !(r_cur & REF_LEAF)) { PT_COUNT_NODE(sc_n) node_visit<QUAD, false>(G, r_cur, r_o, r_inv, r_neg, r_tmax, stk, nn, r_nb3, r_px, r_py, r_pz); }
    if (r_cur != REF_NONE) {
#ifdef PTRS_STEP_COUNTERS
        { const uint32_t k = ((r_cur >> REF_COUNT_SHIFT) & 15u) + 1u; uint32_t kmax = k; for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)kmax, off); kmax = o > kmax ? o : kmax; } PT_COUNT_TRI(sc_n, kmax) } // (an upper bound, diagnostic only)
#endif
        const bool done = leaf_test<false, ALPHA>(G, sc, r_cur, r_o, r_shear, r_tmax, r_h, r_hit, nt, any_rt); r_cur = done ? REF_NONE : pop_next_ref<false>(stk, r_tmax);
    }
}

__device__ inline uint32_t sample_row_of(const DParams &R, uint32_t pid) { // the row of the sample grid a path slot belongs to (path_coord's sy)
    if (R.pixel_mode) return (uint32_t)R.pix_sy;
    const uint32_t npix = (uint32_t)(R.row1 - R.row0) * (uint32_t)R.NX;
    return (uint32_t)R.row0 + (pid % npix) / (uint32_t)R.NX;
}
template <int FEAT>
__device__ inline void epilogue_wave(const DParams &R, const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t kinds_mask, uint32_t seg_cap, uint32_t G, uint32_t s); // below, with k_epilogue
template <int FEAT>
__device__ inline void resolve_wave(const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t seg_cap, uint32_t G, uint32_t s); // below, with k_resolve

// Waves per SIMD the traversal kernels are compiled for.  Quad form (tree in HBM / L2): the kernels wait for node fetches half of their
// wave time (L2 hit rate 0.47-0.57), so a fifth wave pays even where it costs spills -- the full-feature kernels need 110-122 registers
// and take 4-14 spilled ones at 96, almost all in the epilogue / resolve code behind the loop: classroom extend 211 -> 197 ms, connect
// 340 -> 319 ms, frame 923 -> 863 ms.  A sixth wave (80 registers, and only 21 instead of 85 top records in LDS to make room) loses it
// again: colonnade +3 %, classroom +1 %.  LDS form: 6 (the loop needs 61-76 registers); a 10-entry stack column that spares Cornell's tree
// (9 entries) the spill code costs the sixth workgroup per CU and 3 % of the extension kernel.
#ifndef PTRS_QUAD_WAVES
#define PTRS_QUAD_WAVES 5
#endif
#ifndef PTRS_LDS_WAVES
#define PTRS_LDS_WAVES 6
#endif
template <int FEAT, int DEPTH, int GEOM> struct TravWaves { enum { N = (GEOM == 640 || GEOM == 544) ? PTRS_LDS_WAVES : ((GEOM == 0 && DEPTH == 8) ? PTRS_QUAD_WAVES : 0) }; };
// (GEOM = 544 with a 9-entry column: a small scene whose pair tree is one level deeper than the 8-entry column -- Cornell: 527 vectors of LDS form, depth 9 --
// gets the column it needs and a staging area cut to fit: 18 432 + 8 704 B, six workgroups per CU like the 8 / 640 form, and kernels WITHOUT overflow code, whose
// node visit pushes in select form (lf_node_visit).  Alone the 9 / 544 form measured within noise of 8 / 640 + overflow; with the select-form push the
// four-lane Cornell frame went 142.96 -> 139.22 ms (ABAB x 3), the single-lane kernels unchanged: fewer scalar instructions to share the SIMDs' scalar port among
// the lanes' kernels.)
template <int DEPTH, int GEOM> struct TravLds { enum { TOP = GEOM == 0 && DEPTH == 8, V4 = GEOM > 0 ? GEOM : (TOP ? TOP_LDS_STRIDE * QUAD_TOP_NODES : 1) }; }; // quad form with the small stack column: the tree's top lives in LDS

// Diagnostic builds only (-DPTRS_STAMPS_EXT, tools/ablate.sh + tools/stamps_ext.py): wave clocks of the phases of the extension stage, summed
// into Q.stats[CNT_STAMP0 + k] -- 0 retire + bookkeeping, 1 refill (ray loads until the rays are set up), 2 traversal steps, 3 epilogue,
// 4 tickets + segment count, 5 kernel prologue (tables into LDS) -- and counts: 6 wave-steps, 7 refill batches, 8 rays, 9 lanes with a ray
// summed over the wave-steps, 10 segments, 11 lanes at a NODE summed over the wave-steps.  A stamp drains the wave's memory operations first,
// so a phase owns its own latencies.  The product build compiles none of it.
#ifdef PTRS_STAMPS_EXT
#define PT_XS_PARAMS , unsigned long long *xs, unsigned long long &xs_last
#define PT_XS_ARGS , xs, xs_last
#define PT_XS(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); xs[k] += t_ - xs_last; xs_last = t_; }
#define PT_XS_COUNT(k, v) { xs[k] += (unsigned long long)(v); }
#else
#define PT_XS_PARAMS
#define PT_XS_ARGS
#define PT_XS(k)
#define PT_XS_COUNT(k, v)
#endif
// One queue segment through the extension stage: the wave's refill loop, then (kinds_mask != 0) the segment's epilogue.  Called by
// k_extend_rf for every segment the wave takes and by k_tail for its segment's remaining rounds.
template <int FEAT, int DEPTH, bool OVF, int GEOM, bool VOTE>
__device__ inline void extend_segment(const DParams &R, const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t seg_cap, uint32_t thresh, uint32_t kinds_mask, uint32_t Gn, uint32_t s,
                                      const GeomTop &GG, const LdsGeom &LG, LdsStack<DEPTH, OVF> &stk, uint32_t &nn, uint32_t &nt, StepCount &stepc PT_XS_PARAMS) {
    const uint32_t e0 = s * seg_cap, par = it & 1u; // the segment's rays sit at positions e0 .. e0 + n of the round's ray arrays, in queue order: a refill is one coalesced read, no path slot is looked up
    const uint32_t n = rfl(*seg_count(Q, it, Q_EXT, Gn, s));
    uint32_t cursor = 0; // next entry of the segment: wave-uniform, a scalar register
    bool has = false;    // the lane holds an unfinished ray
    uint32_t e = 0;      // its position
    RF_DECL LF_DECL // (per segment: nothing of a ray is live across the epilogue; quad-form kernels use the RF set, LDS-form kernels the LF set)
    for (;;) {
        if (has && r_cur == REF_NONE) { // retire: the hit record is all that leaves this loop
            u4 v; v.x = hit_pack(r_h.prim, r_h.flags); v.y = f2u(r_h.b0); v.z = f2u(r_h.b1); v.w = f2u(r_h.b2);
            pslot(P.hit, e) = v;
            has = false;
        }
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        PT_XS(0)
        if (cursor < n && n_idle >= thresh) {
            PT_XS_COUNT(7, 1) PT_XS_COUNT(8, (n - cursor < n_idle ? n - cursor : n_idle))
            const uint32_t i = cursor + lanes_below(idle);
            if (!has && i < n) {
                e = e0 + i;
                const v4 ov = pslot(P.ray_o[par], e), dv = pslot(P.ray_d[par], e);
                if (GEOM > 0) LF_START(LG, xyz(ov), xyz(dv), PT_INF) else RF_START(xyz(ov), xyz(dv), PT_INF)
                stk.clear(); has = true;
            }
            cursor += n_idle;
            PT_XS(1)
        }
        if (!__any(has)) break; // every ray of the segment is retired
        do {
            PT_XS_COUNT(6, 1) PT_XS_COUNT(9, __popcll(__ballot(has && r_cur != REF_NONE))) PT_XS_COUNT(11, __popcll(__ballot(has && (int32_t)r_cur >= 0))) // steps until enough lanes are through their rays: only then is there something to retire or refill
            if (GEOM > 0) lf_step<VOTE, (FEAT & FEAT_ALPHA) != 0>(sc, r_cur, l_o, l_inv, l_op, l_sx, l_sy, l_sz, l_ox, l_oy, l_oz, l_neg, l_tri, r_tmax, r_h, r_hit, stk, nn, nt, false, stepc);
            else rf_step<VOTE, true, (FEAT & FEAT_ALPHA) != 0>(GG, sc, r_cur, r_o, r_inv, r_neg, r_nb3, r_px, r_py, r_pz, r_shear, r_tmax, r_h, r_hit, stk, nn, nt, false, stepc);
        // with phase voting the wave goes back to retiring / refilling only when that pays: enough lanes are through their rays
        // (or never had one) to reach the refill threshold, or no lane has a step left.  (Going back for every single ray costs
        // a store instruction and the refill bookkeeping per ray: more than the steps saved.)
        } while (VOTE && __any(has && r_cur != REF_NONE) && (cursor >= n || (uint32_t)__popcll(__ballot(!has || r_cur == REF_NONE)) < thresh));
        PT_XS(2)
    }
    // The segment's epilogue runs here, behind the wave's last ray, unless kinds_mask = 0 leaves it to k_epilogue: the hits it
    // reads were written a moment ago by this wave (L2), and its memory latency hides behind the other waves' traversal
    // instead of filling a kernel of its own.  (The fence orders this wave's hit stores before the loads of other lanes.)
    PT_XS(0)
    if (kinds_mask) { __threadfence_block(); epilogue_wave<FEAT>(R, sc, P, Q, it, kinds_mask, seg_cap, Gn, s); }
    PT_XS(3) PT_XS_COUNT(10, 1)
}

template <int FEAT, int DEPTH, bool OVF, int GEOM, bool VOTE>
__global__ __launch_bounds__(BLOCK, (TravWaves<FEAT, DEPTH, GEOM>::N)) void k_extend_rf(DParams R, DScene sc, StackSpill spill, DPaths P, DQueues Q, uint32_t it, uint32_t seg_cap, uint32_t thresh, uint32_t kinds_mask, uint32_t Gn, uint32_t *ticket) {
    __shared__ unsigned long long lds_stack[DEPTH * BLOCK];
    __shared__ v4 lds_geom[TravLds<DEPTH, GEOM>::V4];
    if (round_is_dead(Q, it)) return;
    LdsGeom LG; LG.root = LG.tri0 = LG.tri_copy = 0;
    const GeomTop GG = stage_top(sc, lds_geom, TravLds<DEPTH, GEOM>::TOP);
    if (GEOM > 0) LG = stage_lds_form(sc, lds_geom);
    __syncthreads(); // the only barrier of the kernel: from here on the four waves of the workgroup are independent
    LdsStack<DEPTH, OVF> stk; stk.init(lds_stack, spill);
    uint32_t nn = 0, nt = 0; StepCount stepc;
#ifdef PTRS_STAMPS_EXT
    unsigned long long xs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, xs_last = __builtin_amdgcn_s_memtime();
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) { PT_XS(4) extend_segment<FEAT, DEPTH, OVF, GEOM, VOTE>(R, sc, P, Q, it, seg_cap, thresh, kinds_mask, Gn, s, GG, LG, stk, nn, nt, stepc PT_XS_ARGS); }
    PT_XS(4)
    if ((threadIdx.x & 63u) == 0) for (int k = 0; k < 12; ++k) atomicAdd(&Q.stats[CNT_STAMP0 + k], xs[k]);
#else
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) extend_segment<FEAT, DEPTH, OVF, GEOM, VOTE>(R, sc, P, Q, it, seg_cap, thresh, kinds_mask, Gn, s, GG, LG, stk, nn, nt, stepc);
#endif
    if (R.counters_on) { atomicAdd(&Q.stats[CNT_NODES], (unsigned long long)nn); atomicAdd(&Q.stats[CNT_TRIS], (unsigned long long)nt); atomicAdd(&Q.stats[CNT_NODE_STEPS], (unsigned long long)stepc.node_steps); atomicAdd(&Q.stats[CNT_NODE_VISITS], (unsigned long long)stepc.node_visits); atomicAdd(&Q.stats[CNT_TRI_STEPS], (unsigned long long)stepc.tri_steps); }
}

// The two scene queries of a pending NEE record with lane refill: a lane walks the shadow ray (any hit), then the MIS
// ray (closest hit) of its record.  Shadow-only records (NEE_PRE) are resolved when their ray retires; for the others the
// answers go to the path state (NEE_OCCLUDED in nee2.w, the MIS hit in `hit`, which the shade stage has consumed by now) and
// resolve_wave turns them into radiance with full waves behind the segment's last ray.
// One queue segment through the connection stage (see extend_segment): the wave's refill loop over the segment's NEE records, then
// (fused_resolve) the MIS resolve of the segment.
template <int FEAT, int DEPTH, bool OVF, int GEOM, bool VOTE>
__device__ inline void connect_segment(const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t seg_cap, uint32_t thresh, uint32_t fused_resolve, uint32_t Gn, uint32_t s,
                                       const GeomTop &GG, const LdsGeom &LG, LdsStack<DEPTH, OVF> &stk, uint32_t &nn, uint32_t &nt, StepCount &stepc) {
    const uint32_t f0 = s * seg_cap; // the segment's records sit at positions f0 .. f0 + n of the NEE arrays, in the order of its queue entries
    const uint32_t n = rfl(*seg_count(Q, it, Q_NEE, Gn, s));
    uint32_t cursor = 0;
    bool has = false, shadow_phase = false, setup = false;
    uint32_t pid = 0, fl = 0, f = 0;
    f3 pre_l = splat3(0.0f); // NEE_PRE records: the path's radiance WITH the record's contribution (both fetched with the ray, added at once: three registers live across the traversal instead of seven), stored if the ray comes through
    RF_DECL LF_DECL // (per segment: nothing of a ray is live across the resolve)
    for (;;) {
        if (has && r_cur == REF_NONE && !setup) { // the ray in flight is done
            if (shadow_phase) {
                if (fl & NEE_PRE) { // a shadow-only record is resolved here: l += beta * nLights * ld unless the ray was blocked (integrator.rs:66-78, 444-446)
                    if (!r_hit) pslot(P.L, pid) = mkv4(pre_l, 0.0f); // (L.w is never anything but generate_item's 0)
                    has = false;
                } else {
                    if (r_hit) reinterpret_cast<uint32_t *>(&pslot(P.nee2, f))[3] |= NEE_OCCLUDED << 24;
                    if (fl & NEE_MIS) { shadow_phase = false; setup = true; } else has = false;
                }
            } else {
                u4 v; v.x = (uint32_t)(r_hit ? r_h.prim : -1); v.y = f2u(r_h.b0); v.z = f2u(r_h.b1); v.w = f2u(r_h.b2);
                pslot(P.nhit, f) = v;
                has = false;
            }
        }
        // Ray setup (five divisions) is the expensive part of taking a ray, and it runs with the lanes that need it only: a lane
        // whose shadow ray is done waits for its MIS ray's setup until the idle and the waiting lanes together reach the refill
        // threshold (or nobody else is working), so that one pass of the setup code serves a batch of lanes, not one or two.
        const unsigned long long idle = __ballot(!has);
        const bool batch = (uint32_t)__popcll(__ballot(!has || setup)) >= thresh || !__any(has && !setup);
        if (batch && cursor < n && idle) {
            const uint32_t i = cursor + lanes_below(idle);
            if (!has && i < n) {
                f = f0 + i;
                const uint32_t entry = pslot(Q.nee, f); // path slot | NEE_Q_* (which rays the record holds: no flags word to load before them)
                pid = entry & NEE_Q_PID;
                fl = ((entry & NEE_Q_SHADOW) ? (uint32_t)NEE_SHADOW : 0u) | ((entry & NEE_Q_MIS) ? (uint32_t)NEE_MIS : 0u) | ((entry & NEE_Q_PRE) ? (uint32_t)NEE_PRE : 0u);
#if defined(PTRS_ABL_CONNECT_ONLY) && PTRS_ABL_CONNECT_ONLY == 1
                fl &= ~(uint32_t)NEE_MIS; // diagnostic build: the shadow rays alone (timing only: wrong radiance)
#elif defined(PTRS_ABL_CONNECT_ONLY) && PTRS_ABL_CONNECT_ONLY == 2
                fl &= ~(uint32_t)NEE_SHADOW; // diagnostic build: the MIS rays alone
#endif
                if (fl & (NEE_SHADOW | NEE_MIS)) { shadow_phase = (fl & NEE_SHADOW) != 0; setup = true; has = true; }
            }
            cursor += (uint32_t)__popcll(idle);
        }
        if (batch && setup) {
            const v4 *po = shadow_phase ? P.sh_o : P.mis_o, *pd = shadow_phase ? P.sh_d : P.mis_d; // one copy of the setup code for both kinds of ray
            const v4 o = pslot(po, f), d = pslot(pd, f);
            if (shadow_phase && (fl & NEE_PRE)) { const f3 c = mk3(d.w, o.w, pslot(P.pre_z, f)); pre_l = xyz(pslot(P.L, pid)) + c; } // shade_item's packing of a shadow-only record; resolve_item's sum
            if (GEOM > 0) LF_START(LG, xyz(o), xyz(d), shadow_phase ? PT_SHADOW_TMAX : PT_INF) else RF_START(xyz(o), xyz(d), shadow_phase ? PT_SHADOW_TMAX : PT_INF)
            stk.clear(); setup = false;
        }
        if (!__any(has)) break;
        do {
            if (GEOM > 0) lf_step<VOTE, (FEAT & FEAT_ALPHA) != 0>(sc, r_cur, l_o, l_inv, l_op, l_sx, l_sy, l_sz, l_ox, l_oy, l_oz, l_neg, l_tri, r_tmax, r_h, r_hit, stk, nn, nt, shadow_phase, stepc);
            else rf_step<VOTE, true, (FEAT & FEAT_ALPHA) != 0>(GG, sc, r_cur, r_o, r_inv, r_neg, r_nb3, r_px, r_py, r_pz, r_shear, r_tmax, r_h, r_hit, stk, nn, nt, shadow_phase, stepc);
        } while (VOTE && __any(has && r_cur != REF_NONE) && (cursor >= n || (uint32_t)__popcll(__ballot(!has || r_cur == REF_NONE)) < thresh)); // (see k_extend_rf)
    }
    // records with a MIS ray are resolved behind the wave's last ray (see k_extend_rf's epilogue), unless fused_resolve = 0 leaves them to k_resolve
    if (fused_resolve) { __threadfence_block(); resolve_wave<FEAT>(sc, P, Q, it, seg_cap, Gn, s); }
}

template <int FEAT, int DEPTH, bool OVF, int GEOM, bool VOTE>
__global__ __launch_bounds__(BLOCK, (TravWaves<FEAT, DEPTH, GEOM>::N)) void k_connect_rf(DParams R, DScene sc, StackSpill spill, DPaths P, DQueues Q, uint32_t it, uint32_t seg_cap, uint32_t thresh, uint32_t fused_resolve, uint32_t Gn, uint32_t *ticket) {
    __shared__ unsigned long long lds_stack[DEPTH * BLOCK];
    __shared__ v4 lds_geom[TravLds<DEPTH, GEOM>::V4];
    if (round_is_dead(Q, it)) return;
    LdsGeom LG; LG.root = LG.tri0 = LG.tri_copy = 0;
    const GeomTop GG = stage_top(sc, lds_geom, TravLds<DEPTH, GEOM>::TOP);
    if (GEOM > 0) LG = stage_lds_form(sc, lds_geom);
    __syncthreads();
    LdsStack<DEPTH, OVF> stk; stk.init(lds_stack, spill);
    uint32_t nn = 0, nt = 0; StepCount stepc;
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) connect_segment<FEAT, DEPTH, OVF, GEOM, VOTE>(sc, P, Q, it, seg_cap, thresh, fused_resolve, Gn, s, GG, LG, stk, nn, nt, stepc);
    if (R.counters_on) { atomicAdd(&Q.stats[CNT_NODES], (unsigned long long)nn); atomicAdd(&Q.stats[CNT_TRIS], (unsigned long long)nt); atomicAdd(&Q.stats[CNT_NODE_STEPS], (unsigned long long)stepc.node_steps); atomicAdd(&Q.stats[CNT_NODE_VISITS], (unsigned long long)stepc.node_visits); atomicAdd(&Q.stats[CNT_TRI_STEPS], (unsigned long long)stepc.tri_steps); }
}

// estimate_direct's use of the two answers (integrator.rs:66-78, 121-134) and `l += beta * nLights * ld`, full waves
template <int FEAT>
__device__ inline void resolve_wave(const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t seg_cap, uint32_t G, uint32_t s) {
    const uint32_t f0 = s * seg_cap;
    const uint32_t n = rfl(*seg_count(Q, it, Q_NEE, G, s));
    if (rfl(*seg_count(Q, it, Q_MIS, G, s)) == 0) return; // every record of the segment was shadow-only: k_connect_rf has resolved them
    for (uint32_t i = __lane_id(); i < n; i += 64u) {
        const uint32_t f = f0 + i;
        const uint32_t entry = pslot(Q.nee, f);
        if (entry & NEE_Q_PRE) continue; // shadow-only: resolved when its ray retired
        const uint32_t fl = pslot(P.nee2, f).w >> 24;
        HitRec mh; mh.prim = -1; mh.t = 0.0f; mh.b0 = mh.b1 = mh.b2 = 0.0f; mh.flags = 0;
        if (fl & NEE_MIS) { const u4 v = pslot(P.nhit, f); mh.prim = (int32_t)v.x; mh.b0 = u2f(v.y); mh.b1 = u2f(v.z); mh.b2 = u2f(v.w); }
        resolve_item<FEAT>(sc, P, entry, f, (fl & NEE_OCCLUDED) != 0, mh);
    }
}
template <int FEAT>
__global__ __launch_bounds__(BLOCK) void k_resolve(DScene sc, DPaths P, DQueues Q, uint32_t it, uint32_t seg_cap, uint32_t Gn, uint32_t *ticket) {
    if (round_is_dead(Q, it)) return;
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) resolve_wave<FEAT>(sc, P, Q, it, seg_cap, Gn, s);
}

// The epilogue of one queue segment, full waves: emission / environment / depth cut (integrator.rs:418-431) and material
// bucketing, in queue order: hits, state words and path slots of 64 consecutive positions per step.
template <int FEAT>
__device__ inline void epilogue_wave(const DParams &R, const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t kinds_mask, uint32_t seg_cap, uint32_t G, uint32_t s) {
    const uint32_t lane = __lane_id();
    const uint32_t e0 = s * seg_cap, par = it & 1u;
    const uint32_t n = rfl(*seg_count(Q, it, Q_EXT, G, s));
    uint32_t cnt[6] = {0u, 0u, 0u, 0u, 0u, 0u}; // wave-uniform: scalar registers
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const uint32_t i = i0 + lane, e = e0 + i;
        int k = -1; uint32_t pid = 0;
        if (i < n) {
            pid = pslot(Q.ext[par], e);
            if (R.row_cost) atomicAdd(R.row_cost + sample_row_of(R, pid), 1u); // (band planning probe only: the extension ray this path traced this round)
            const u4 r = pslot(P.hit, e); // (the hits, the state words and the path slots of 64 consecutive positions: three coalesced reads)
            HitRec h; h.prim = hit_prim(r.x); h.t = 0.0f; h.b0 = u2f(r.y); h.b1 = u2f(r.z); h.b2 = u2f(r.w);
            h.flags = hit_flags(r.x); // the two fields of the leaf record's flags the epilogue reads
            k = extension_epilogue<FEAT>(R, sc, P, par, pid, e, h);
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) { // wavefront-ballot bucketing by material kind
            if (!(kinds_mask & (1u << m))) continue;
            const uint32_t slot = wave_push(cnt[m], k == m);
            if (k == m) { MatEntry me; me.e = e; me.pid = pid; pslot(Q.mat[m], e0 + slot) = me; }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < 6; ++m) if (kinds_mask & (1u << m)) *seg_count(Q, it, Q_MAT0 + m, G, s) = cnt[m];
    }
}

// The epilogue k_extend_rf leaves out when it does not run it itself (option fused_epilogue = 0).
template <int FEAT>
__global__ __launch_bounds__(BLOCK) void k_epilogue(DParams R, DScene sc, DPaths P, DQueues Q, uint32_t it, uint32_t kinds_mask, uint32_t seg_cap, uint32_t Gn, uint32_t *ticket) {
    if (round_is_dead(Q, it)) return;
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) epilogue_wave<FEAT>(R, sc, P, Q, it, kinds_mask, seg_cap, Gn, s);
}


// One instantiation per material kind (and feature set): lobe kinds are compile-time constants.
// Occupancy hint (waves per SIMD) per instantiation, from A/B runs on MI355X.  The Matte / FEAT_SIMPLE kernel needs 197 registers:
// round 2 ran it at 3 waves (168 registers, 24 bytes of scratch, +1.6 % on the Cornell frame); with this round's wave-per-segment
// loop around it 168 registers cost 24 spilled ones (116 bytes of scratch, each reload a trip to L1 / L2 that 2.7 waves per SIMD do
// not cover), and at 2 waves without spills the kernel is 2 % faster alone and the three-lane frame 5 % (177.9-179.2 -> 169.3-169.8
// ms: its workgroups leave room for the other lanes' traversal kernels).  The mirror / glass kernels need 109-135 registers and get
// their third wave from the LDS budget alone; the Disney kernel with image textures (colonnade) needs 209: at 168 with 68 spilled registers it was
// faster alone on one lane (shade kernels 42.3 -> 39.6 ms), but with four lanes sharing the machine the 2-wave kernel without scratch wins
// (colonnade 113.2-113.6 -> 109.9-110.0 ms: no scratch set-up per dispatch, room for the other lanes' traversal waves); the other
// Disney / metal / substrate kernels (191-256 registers) stay at 2.
#ifndef PTRS_SHADE_WAVES_MATTE
#define PTRS_SHADE_WAVES_MATTE 2
#endif
#ifndef PTRS_SHADE_WAVES_DISNEY_IMG
#define PTRS_SHADE_WAVES_DISNEY_IMG 2
#endif
template <int MAT, int FEAT> struct ShadeWaves { enum { N = (MAT == 0 && FEAT == FEAT_SIMPLE) ? PTRS_SHADE_WAVES_MATTE : ((MAT == 4 && FEAT == FEAT_IMG) ? PTRS_SHADE_WAVES_DISNEY_IMG : 2) }; }; // 2: never above 256 registers (one wave per SIMD otherwise)

// The shade kernels' read-only tables in LDS.  A shading vertex issues ~150 vector-memory instructions -- path state,
// the triangle's record, 32 Sobol' table words, the light's record, spills -- and the kernel's time is the time the
// CU's address unit needs for them (TA busy 70 %, 37 busy cycles per instruction; VALU half idle; HBM at a fifth of its
// rate).  What is small and shared by all paths therefore moves into LDS once per workgroup:
//   * the light records (<= SH_LIGHTS of them),
//   * the triangles' shading records when the scene has <= SH_TRI_V4 / 13 triangles (Cornell: 36),
//   * the Sobol' tables of the dimensions this round can draw, in nibble form (16 words per index nibble and dimension):
//     a sample is scramble ^ XOR of 8 (32-bit index) or 13 LDS words instead of 4 or 8 words gathered from L2.
#ifndef PTRS_SH_SOB_WORDS
#define PTRS_SH_SOB_WORDS 3200
#endif
#ifndef PTRS_SH_TRI_V4
#define PTRS_SH_TRI_V4 512
#endif
enum : uint32_t { SH_SOB_WORDS = PTRS_SH_SOB_WORDS, SH_TRI_V4 = PTRS_SH_TRI_V4, SH_LIGHTS = 16, SH_LIGHT_V4 = sizeof(DLight) / 16, SH_TRI_REC_V4 = sizeof(DTriShade) / 16 };
static_assert(sizeof(DLight) % 16 == 0 && sizeof(DTriShade) % 16 == 0, "records are staged as 16-byte vectors");
struct ShadeLdsCfg { uint32_t sob_lo, sob_n, sob_nib, tri_lds, n_lights_lds, marg_li, pre_li; }; // pre_li: the environment light whose samples k_env_presample has left in pre0 / pre1 (0xffffffff: none) // marg_li: the environment light whose marginal tables are staged (0xffffffff: none)
enum : uint32_t { SH_MARG_N = 1024, SH_MARG_WORDS = 3 * SH_MARG_N + 8 }; // row integrals [nv] | their cdf [nv + 1] | the cdf's guide [guide_v + 1], nv and guide_v <= 1024 // Sobol' window [sob_lo, sob_lo + sob_n), nibbles staged per dimension
typedef __attribute__((address_space(3))) const uint32_t lds_u32;
PT_HD uint32_t sob_stride(uint32_t nib) { return nib * 16u + 4u; } // words per dimension; + 4: consecutive dimensions start 4 banks apart
template <bool ENVPRE = false>
struct ShadeCtxLds {
    static constexpr bool inf_fallback = !ENVPRE; // ENVPRE: the scene's one InfiniteAreaLight is presampled (k_env_presample), shade_item never walks its distribution itself
    lds_u32 *sob; lds_v4 *tris, *lights; ShadeLdsCfg cfg; InfMarginal marg;
    __device__ inline const InfMarginal *inf_marginal(uint32_t li) const { return li == cfg.marg_li ? &marg : nullptr; }
    __device__ inline bool presampled(uint32_t li) const { return li == cfg.pre_li; }
    __device__ inline v4 ld(lds_v4 *q) const { v4 r; r.x = q->x; r.y = q->y; r.z = q->z; r.w = q->w; return r; }
    __device__ inline TriRegs tri(const DScene &sc, int32_t prim, bool want_dp) const {
        if (!cfg.tri_lds) return load_tri_regs(sc.shade + prim, want_dp);
        v4 r[SH_TRI_REC_V4];
        lds_v4 *q = tris + (uint32_t)prim * SH_TRI_REC_V4;
#pragma unroll
        for (uint32_t k = 0; k < SH_TRI_REC_V4; ++k) r[k] = ld(q + k);
        return load_tri_regs(reinterpret_cast<const DTriShade *>(r), want_dp);
    }
    __device__ inline void before_stores() const { __builtin_amdgcn_s_waitcnt(0x0F70); } // vmcnt(0): the next item's state has landed in LDS (k_shade)
    __device__ inline void light(const DScene &sc, uint32_t li, DLight &out) const {
        if (li >= cfg.n_lights_lds) { out = sc.lights[li]; return; }
        v4 *o = reinterpret_cast<v4 *>(&out);
        lds_v4 *q = lights + li * SH_LIGHT_V4;
#pragma unroll
        for (uint32_t k = 0; k < SH_LIGHT_V4; ++k) o[k] = ld(q + k);
    }
    template <int N> __device__ inline void sobol(const DSampler &S, uint64_t index, const uint32_t (&dim)[N], uint32_t scramble, float (&out)[N]) const {
        const uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
        bool in = cfg.sob_nib >= 13u || (hi == 0u && cfg.sob_nib >= 8u); // every set bit of the index has its nibble staged
#pragma unroll
        for (int k = 0; k < N; ++k) in = in && dim[k] - cfg.sob_lo < cfg.sob_n;
        if (!in) { sobol_batch<N>(S, index, dim, scramble, out); return; } // outside this round's window: the global tables
        uint32_t v[N];
        bool run = cfg.sob_nib == 8u; // the common case: N consecutive dimensions, 8 nibbles staged
#pragma unroll
        for (int k = 1; k < N; ++k) run = run && dim[k] == dim[0] + (uint32_t)k;
        if (run) { // one address per index nibble, the dimension is a compile-time offset of the LDS read
            constexpr uint32_t ST = 8u * 16u + 4u; // sob_stride(8)
            lds_u32 *t = sob + (dim[0] - cfg.sob_lo) * ST;
            lds_u32 *t0 = t + (lo & 15u), *t1 = t + 16u + ((lo >> 4) & 15u), *t2 = t + 32u + ((lo >> 8) & 15u), *t3 = t + 48u + ((lo >> 12) & 15u), *t4 = t + 64u + ((lo >> 16) & 15u),
                    *t5 = t + 80u + ((lo >> 20) & 15u), *t6 = t + 96u + ((lo >> 24) & 15u), *t7 = t + 112u + (lo >> 28);
#pragma unroll
            for (int k = 0; k < N; ++k) v[k] = scramble ^ t0[k * ST] ^ t1[k * ST] ^ t2[k * ST] ^ t3[k * ST] ^ t4[k * ST] ^ t5[k * ST] ^ t6[k * ST] ^ t7[k * ST];
#pragma unroll
            for (int k = 0; k < N; ++k) out[k] = min_nz(PT_ONE_MINUS_EPS, (float)v[k] * 0x1p-32f);
            return;
        }
        const uint32_t stride = sob_stride(cfg.sob_nib);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            lds_u32 *t = sob + (dim[k] - cfg.sob_lo) * stride;
            v[k] = scramble ^ t[lo & 15u] ^ t[16u + ((lo >> 4) & 15u)] ^ t[32u + ((lo >> 8) & 15u)] ^ t[48u + ((lo >> 12) & 15u)] ^ t[64u + ((lo >> 16) & 15u)] ^ t[80u + ((lo >> 20) & 15u)] ^
                   t[96u + ((lo >> 24) & 15u)] ^ t[112u + (lo >> 28)];
        }
        if (cfg.sob_nib > 8u) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                lds_u32 *t = sob + (dim[k] - cfg.sob_lo) * stride;
                v[k] ^= t[128u + (hi & 15u)] ^ t[144u + ((hi >> 4) & 15u)] ^ t[160u + ((hi >> 8) & 15u)] ^ t[176u + ((hi >> 12) & 15u)] ^ t[192u + ((hi >> 16) & 15u)];
            }
        }
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = min_nz(PT_ONE_MINUS_EPS, (float)v[k] * 0x1p-32f);
    }
};

// The shade kernels' LDS: the next item's path state per thread (filled by LDS-DMA) and the read-only tables.
template <int FEAT, bool ENVPRE> struct ShadeLdsSizes { enum : uint32_t { NPF = (FEAT & FEAT_INFINITE) ? 7u : 5u /* + the vertex's presampled environment-light sample */, MARG = ((FEAT & FEAT_INFINITE) && !ENVPRE) ? 1u : 0u /* the in-kernel walk of the environment light's distribution needs its marginal tables */ }; };
// Staging of the tables, once per workgroup (the caller's barrier makes them visible); sob_n = 0 stages no Sobol' window (k_tail:
// its waves are in different rounds, their draws read the global tables).
template <int FEAT, bool ENVPRE>
__device__ inline ShadeCtxLds<ENVPRE> stage_shade_tables(const DSampler &S, const DScene &sc, const ShadeLdsCfg &cfg, uint32_t *lds_sob, v4 *lds_tri, v4 *lds_light, float *lds_marg) {
    constexpr bool MARG = ShadeLdsSizes<FEAT, ENVPRE>::MARG != 0;
    {
        const uint32_t stride = sob_stride(cfg.sob_nib), nw = cfg.sob_nib * 16u;
        for (uint32_t i = threadIdx.x; i < cfg.sob_n * stride; i += BLOCK) { const uint32_t d = i / stride, w = i - d * stride; lds_sob[i] = w < nw ? S.nibtab[((size_t)(cfg.sob_lo + d) * SOBOL_NIBBLES) * 16u + w] : 0u; }
        if (cfg.tri_lds) { const v4 *g = reinterpret_cast<const v4 *>(sc.shade); for (uint32_t i = threadIdx.x; i < sc.n_prims * SH_TRI_REC_V4; i += BLOCK) lds_tri[i] = g[i]; }
        const v4 *gl = reinterpret_cast<const v4 *>(sc.lights);
        for (uint32_t i = threadIdx.x; i < cfg.n_lights_lds * SH_LIGHT_V4; i += BLOCK) lds_light[i] = gl[i];
        if (MARG && cfg.marg_li != 0xffffffffu) { // the environment light's marginal distribution: 12 KB that every light sample walks first
            const DLight &Le = sc.lights[cfg.marg_li];
            const uint32_t nv = (uint32_t)Le.nv, gv = Le.guide_v;
            for (uint32_t i = threadIdx.x; i < nv; i += BLOCK) lds_marg[i] = sc.distdata[Le.fint_off + i];
            for (uint32_t i = threadIdx.x; i < nv + 1u; i += BLOCK) lds_marg[SH_MARG_N + i] = sc.distdata[Le.mcdf_off + i];
            for (uint32_t i = threadIdx.x; i < gv + 1u; i += BLOCK) lds_marg[2u * SH_MARG_N + 1u + i] = sc.distdata[Le.mguide_off + i];
        }
    }
    ShadeCtxLds<ENVPRE> X; X.sob = (lds_u32 *)lds_sob; X.tris = (lds_v4 *)lds_tri; X.lights = (lds_v4 *)lds_light; X.cfg = cfg;
    if (!MARG) X.cfg.marg_li = 0xffffffffu;
    X.marg.func = lds_marg; X.marg.cdf = lds_marg + (MARG ? SH_MARG_N : 0); X.marg.guide = lds_marg + (MARG ? 2 * SH_MARG_N + 1 : 0);
    return X;
}

// One queue segment of one material bucket through the shade stage.  Called by k_shade for every segment the wave takes and by
// k_tail for its segment's remaining rounds.
// Software pipeline through LDS.  A vertex needs 80 bytes of its path out of HBM before it can start, and nothing else
// of this stage waits for HBM: the wave asks for the NEXT item's five vectors with LDS-DMA loads (global_load_lds_dwordx4:
// no registers held while they fly) before it shades the current item, and shade_item waits for them right before it issues
// its stores (before_stores) -- by then they have had a whole vertex's time to arrive.  The queue entry is read two items
// ahead.  Vector k of wave w sits at lds_pf[(k * 4 + w) * 64 + lane].
template <int MAT, int FEAT, bool ENVPRE>
__device__ inline void shade_segment(const DParams &R, const DSampler &S, const DCamera &C, const DScene &sc, const DPaths &P, const DQueues &Q, uint32_t it, uint32_t seg_cap, uint32_t Gn, uint32_t s,
                                     const ShadeCtxLds<ENVPRE> &X, v4 *lds_pf, bool &err_dim PT_STAMP_PARAMS) {
    const ShadeLdsCfg &cfg = X.cfg;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u, par = it & 1u;
    auto dma = [&](const MatEntry &m) { // the vertex's ray, throughput and hit at its position in the round's extension queue, the path's constants at its slot
        typedef __attribute__((address_space(1))) const void gptr; typedef __attribute__((address_space(3))) void lptr;
        if (FEAT & FEAT_IMAGE) __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.ray_o[par], m.e), (lptr *)(lds_pf + (0u * 4u + wv) * 64u), 16, 0, 0); // (the ray's origin is only read for the camera ray's differentials, which only image-texture lookups use)
        __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.ray_d[par], m.e), (lptr *)(lds_pf + (1u * 4u + wv) * 64u), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.beta[par], m.e), (lptr *)(lds_pf + (2u * 4u + wv) * 64u), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.st, m.pid), (lptr *)(lds_pf + (3u * 4u + wv) * 64u), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.hit, m.e), (lptr *)(lds_pf + (4u * 4u + wv) * 64u), 16, 0, 0);
        if ((FEAT & FEAT_INFINITE) && cfg.pre_li != 0xffffffffu) {
            __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.pre0, m.e), (lptr *)(lds_pf + (5u * 4u + wv) * 64u), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr *)&pslot(P.pre1, m.e), (lptr *)(lds_pf + (6u * 4u + wv) * 64u), 16, 0, 0);
        }
    };
    const MatEntry *__restrict__ queue = Q.mat[MAT] + (size_t)s * seg_cap;
    const uint32_t n = rfl(*seg_count(Q, it, Q_MAT0 + MAT, Gn, s));
    if (n == 0) return;
    // several material kernels append to the same output segments one after the other
    const uint32_t next_base = rfl(*seg_count(Q, it + 1u, Q_EXT, Gn, s)), nee_base = rfl(*seg_count(Q, it, Q_NEE, Gn, s));
    const uint32_t e_next0 = s * seg_cap + next_base, f0 = s * seg_cap + nee_base; // where this launch's first continuing ray / first NEE record goes
    uint32_t c_next = 0, c_nee = 0, c_shadow = 0, c_mis = 0; // the segment's output counters: wave-uniform, scalar registers
    uint32_t i = lane;
    MatEntry m, m1; m.e = m.pid = m1.e = m1.pid = 0;
    if (i < n) { m = pslot(queue, i); dma(m); }
    if (i + 64u < n) m1 = pslot(queue, i + 64u);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the first item's state is in LDS
    while (i < n) {
        PathIn in;
        {
            lds_v4 *q = (lds_v4 *)lds_pf + threadIdx.x;
            if (FEAT & FEAT_IMAGE) in.ro = X.ld(q); else in.ro = mkv4(splat3(0.0f), 0.0f);
            in.rd = X.ld(q + BLOCK); in.beta = X.ld(q + 2 * BLOCK);
            const v4 a = X.ld(q + 3 * BLOCK), c = X.ld(q + 4 * BLOCK);
            in.st.x = f2u(a.x); in.st.y = f2u(a.y); in.st.z = f2u(a.z); in.st.w = f2u(a.w);
            in.hit.x = f2u(c.x); in.hit.y = f2u(c.y); in.hit.z = f2u(c.z); in.hit.w = f2u(c.w);
            if ((FEAT & FEAT_INFINITE) && cfg.pre_li != 0xffffffffu) { in.pre0 = X.ld(q + 5 * BLOCK); in.pre1 = X.ld(q + 6 * BLOCK); } else in.pre0 = in.pre1 = mkv4(splat3(0.0f), 0.0f);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the LDS reads are done before the next DMA overwrites the buffer
        const uint32_t i2 = i + 64u;
        if (i2 < n) dma(m1);
        MatEntry m2; m2.e = m2.pid = 0;
        if (i2 + 64u < n) m2 = pslot(queue, i2 + 64u);
#ifdef PTRS_STAMPS
        const ShadeResult r = shade_item<MAT, FEAT>(R, S, C, sc, P, m.pid, in, X, stamp_acc, stamp_last);
#else
        const ShadeResult r = shade_item<MAT, FEAT>(R, S, C, sc, P, m.pid, in, X);
#endif
        err_dim = err_dim || r.err_dim;
        // the vertex's stores: the continuing ray at the position the path takes in the next round's extension queue, the NEE record at
        // the position of its queue entry -- consecutive positions for the lanes of the wave, whole 128-byte lines
        const uint32_t e_next = e_next0 + wave_push(c_next, r.next), f = f0 + wave_push(c_nee, r.nee);
        X.before_stores();
        store_shade_out(P, par ^ 1u, r, e_next, f);
        if (r.next) pslot(Q.ext[par ^ 1u], e_next) = m.pid;
        if (r.nee) pslot(Q.nee, f) = r.nee_entry(m.pid);
#ifdef PTRS_STAMPS
        { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[8] += t_ - stamp_last; stamp_last = t_; stamp_acc[10] += 1; }
#endif
        c_shadow += (uint32_t)__popcll(__ballot(r.shadow));
        c_mis += (uint32_t)__popcll(__ballot(r.mis));
        if (R.row_cost && (r.shadow || r.mis)) atomicAdd(R.row_cost + sample_row_of(R, m.pid), (r.shadow ? 1u : 0u) + (r.mis ? 1u : 0u)); // (band planning probe only)
        i = i2; m = m1; m1 = m2;
    }
    if (lane == 0) { // the wave is the segment's only writer in this launch
        *seg_count(Q, it + 1u, Q_EXT, Gn, s) = next_base + c_next;
        if (c_next) Q.alive[it + 1u] = 1u;
        *seg_count(Q, it, Q_NEE, Gn, s) = nee_base + c_nee;
        *seg_count(Q, it, Q_SHADOW, Gn, s) += c_shadow;
        *seg_count(Q, it, Q_MIS, Gn, s) += c_mis;
    }
}

template <int MAT, int FEAT, bool ENVPRE>
__global__ __launch_bounds__(BLOCK, (ShadeWaves<MAT, FEAT>::N)) void k_shade(DParams R, DSampler S, DCamera C, DScene sc, DPaths P, DQueues Q, uint32_t it, uint32_t seg_cap, ShadeLdsCfg cfg, uint32_t Gn, uint32_t *ticket) {
    __shared__ v4 lds_pf[ShadeLdsSizes<FEAT, ENVPRE>::NPF * BLOCK]; // the next item's path state per thread, filled by LDS-DMA
    __shared__ uint32_t lds_sob[SH_SOB_WORDS];
    __shared__ v4 lds_tri[SH_TRI_V4];
    __shared__ v4 lds_light[SH_LIGHTS * SH_LIGHT_V4];
    __shared__ float lds_marg[ShadeLdsSizes<FEAT, ENVPRE>::MARG ? SH_MARG_WORDS : 1];
    bool err_dim = false;
    if (round_is_dead(Q, it)) return;
    const ShadeCtxLds<ENVPRE> X = stage_shade_tables<FEAT, ENVPRE>(S, sc, cfg, lds_sob, lds_tri, lds_light, lds_marg);
    __syncthreads(); // the only barrier: the tables are staged once per workgroup, then its four waves work through segments on their own
#ifdef PTRS_STAMPS
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) shade_segment<MAT, FEAT, ENVPRE>(R, S, C, sc, P, Q, it, seg_cap, Gn, s, X, lds_pf, err_dim, stamp_acc, stamp_last);
#else
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) shade_segment<MAT, FEAT, ENVPRE>(R, S, C, sc, P, Q, it, seg_cap, Gn, s, X, lds_pf, err_dim);
#endif
    if (err_dim) atomicOr(&Q.stats[CNT_ERR], (unsigned long long)PTRS_ERRFLAG_SOBOL_DIM);
#ifdef PTRS_STAMPS
    { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[11] += t_ - stamp_last; }
    if ((threadIdx.x & 63u) == 0) for (int k = 0; k < 12; ++k) atomicAdd(&Q.stats[CNT_STAMP0 + k], stamp_acc[k]);
#endif
}

// ---- the fused tail: a wave takes ITS segment through all remaining rounds --------------------------------------------------------
// A pass is max_depth + 1 rounds of extend -> shade -> connect, three dependent launches each; after the first few rounds Russian
// roulette and the depth of the scene have left a handful of paths per segment, and every launch of such a round costs 30-60 us
// whatever it holds (dispatch, tables into LDS, tickets, the drain): ~7 ms per render call that do not shrink with the job
// (tools/fixed_cost_probe.py), which is what caps the strong scaling of a frame over N GPUs.  But segments are wave-private through
// every stage -- the wave that consumes segment s of one stage appends only to segment s of the next -- so once a pass is thin no
// grid-wide ordering is needed at all: ONE launch in which each wave runs extend -> shade -> connect for its own segment, round
// after round, until the segment is empty.  Same stage functions (extend_segment, shade_segment, connect_segment) on the same
// queue positions in the same order: same bits.  A workgroup-scope fence between the stages orders the wave's stores before its
// next stage's loads (what epilogue_wave / resolve_wave rely on already).  The Sobol' window of the shade stage is per round and
// the waves of a workgroup are in different rounds: draws read the global tables (sob_n = 0).
// Instantiated for scenes of ONE material bucket whose traversal kernels are the lean ones (Cornell: Matte; colonnade: Disney +
// image textures); two workgroups per CU (the Matte shade stage's registers, the LDS of both stages).
template <int MAT, int FEAT, int GEOM, bool OVF, int DEPTH = 8>
__global__ __launch_bounds__(BLOCK, 2) void k_tail(DParams R, DSampler S, DCamera C, DScene sc, StackSpill spill, DPaths P, DQueues Q, uint32_t it0, uint32_t it_end, uint32_t seg_cap, uint32_t thresh_e, uint32_t thresh_c,
                                                   uint32_t kinds_mask, ShadeLdsCfg cfg, uint32_t Gn, uint32_t *ticket) {
    constexpr int FEAT_T = FEAT_SIMPLE;
    constexpr bool VOTE_C = GEOM == 0; // (the launch policy's defaults: phase voting in the connection stage for quad-form scenes only)
    __shared__ unsigned long long lds_stack[DEPTH * BLOCK];
    __shared__ v4 lds_geom[TravLds<DEPTH, GEOM>::V4];
    __shared__ v4 lds_pf[ShadeLdsSizes<FEAT, false>::NPF * BLOCK];
    __shared__ uint32_t lds_sob[4];
    __shared__ v4 lds_tri[SH_TRI_V4];
    __shared__ v4 lds_light[SH_LIGHTS * SH_LIGHT_V4];
    __shared__ float lds_marg[ShadeLdsSizes<FEAT, false>::MARG ? SH_MARG_WORDS : 1];
    if (round_is_dead(Q, it0)) return;
    LdsGeom LG; LG.root = LG.tri0 = LG.tri_copy = 0;
    const GeomTop GG = stage_top(sc, lds_geom, TravLds<DEPTH, GEOM>::TOP);
    if (GEOM > 0) LG = stage_lds_form(sc, lds_geom);
    cfg.sob_n = 0;
    const ShadeCtxLds<false> X = stage_shade_tables<FEAT, false>(S, sc, cfg, lds_sob, lds_tri, lds_light, lds_marg);
    __syncthreads(); // the only barrier
    LdsStack<DEPTH, OVF> stk; stk.init(lds_stack, spill);
    uint32_t nn = 0, nt = 0; StepCount stepc;
    bool err_dim = false;
#ifdef PTRS_STAMPS
    unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) {
        for (uint32_t it = it0; it < it_end; ++it) {
            if (rfl(*(volatile uint32_t *)seg_count(Q, it, Q_EXT, Gn, s)) == 0u) break; // nobody of this segment reached the round
#ifdef PTRS_STAMPS_EXT
            { unsigned long long xs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, xs_last = 0; extend_segment<FEAT_T, DEPTH, OVF, GEOM, true>(R, sc, P, Q, it, seg_cap, thresh_e, kinds_mask, Gn, s, GG, LG, stk, nn, nt, stepc PT_XS_ARGS); }
#else
            extend_segment<FEAT_T, DEPTH, OVF, GEOM, true>(R, sc, P, Q, it, seg_cap, thresh_e, kinds_mask, Gn, s, GG, LG, stk, nn, nt, stepc);
#endif
            __threadfence_block();
#ifdef PTRS_STAMPS
            shade_segment<MAT, FEAT, false>(R, S, C, sc, P, Q, it, seg_cap, Gn, s, X, lds_pf, err_dim, stamp_acc, stamp_last);
#else
            shade_segment<MAT, FEAT, false>(R, S, C, sc, P, Q, it, seg_cap, Gn, s, X, lds_pf, err_dim);
#endif
            __threadfence_block();
            connect_segment<FEAT_T, DEPTH, OVF, GEOM, VOTE_C>(sc, P, Q, it, seg_cap, thresh_c, 1u, Gn, s, GG, LG, stk, nn, nt, stepc);
            __threadfence_block();
        }
    }
    if (err_dim) atomicOr(&Q.stats[CNT_ERR], (unsigned long long)PTRS_ERRFLAG_SOBOL_DIM);
    if (R.counters_on) { atomicAdd(&Q.stats[CNT_NODES], (unsigned long long)nn); atomicAdd(&Q.stats[CNT_TRIS], (unsigned long long)nt); atomicAdd(&Q.stats[CNT_NODE_STEPS], (unsigned long long)stepc.node_steps); atomicAdd(&Q.stats[CNT_NODE_VISITS], (unsigned long long)stepc.node_visits); atomicAdd(&Q.stats[CNT_TRI_STEPS], (unsigned long long)stepc.tri_steps); }
}

// The environment light's samples of a round's vertices, ahead of the round's shade kernels.  Which light a vertex samples and with
// which numbers is settled by its Sobol' draws alone (light.rs:402-441, integrator.rs:192-217), and for an InfiniteAreaLight the sample's
// direction, pdf and radiance do not depend on the shading point: a chain of ~8 dependent reads (the marginal walk, the row's guide and
// cdf, the map's texels) that the shade kernels used to make at 2 waves per SIMD with 250 registers live around it.  Here it runs
// with a third of the registers and the latency of one vertex hidden behind the others'; the shade kernel receives (wi, pdf | Li, valid)
// in pre0 / pre1 (by the vertex's position in the extension queue) with the rest of the vertex's state.
template <int FEAT>
__global__ __launch_bounds__(BLOCK) void k_env_presample(DSampler S, DScene sc, DPaths P, DQueues Q, uint32_t it, uint32_t seg_cap, uint32_t nee_kinds, uint32_t env_li, ShadeLdsCfg cfg, uint32_t Gn, uint32_t *ticket) {
    __shared__ float lds_marg[SH_MARG_WORDS];
    __shared__ uint32_t lds_sob[SH_SOB_WORDS]; // the round's Sobol' window in nibble form, as in k_shade (from the global byte tables a draw is four 4-byte gathers out of L2: 12 per vertex here, which made this kernel cost what it saves)
    if (round_is_dead(Q, it)) return;
    const DLight &Le = sc.lights[env_li];
    {
        const uint32_t stride = sob_stride(cfg.sob_nib), nw = cfg.sob_nib * 16u;
        for (uint32_t i = threadIdx.x; i < cfg.sob_n * stride; i += BLOCK) { const uint32_t d = i / stride, w = i - d * stride; lds_sob[i] = w < nw ? S.nibtab[((size_t)(cfg.sob_lo + d) * SOBOL_NIBBLES) * 16u + w] : 0u; }
        const uint32_t nv = (uint32_t)Le.nv, gv = Le.guide_v;
        for (uint32_t i = threadIdx.x; i < nv; i += BLOCK) lds_marg[i] = sc.distdata[Le.fint_off + i];
        for (uint32_t i = threadIdx.x; i < nv + 1u; i += BLOCK) lds_marg[SH_MARG_N + i] = sc.distdata[Le.mcdf_off + i];
        for (uint32_t i = threadIdx.x; i < gv + 1u; i += BLOCK) lds_marg[2u * SH_MARG_N + 1u + i] = sc.distdata[Le.mguide_off + i];
    }
    __syncthreads();
    InfMarginal marg; marg.func = lds_marg; marg.cdf = lds_marg + SH_MARG_N; marg.guide = lds_marg + 2 * SH_MARG_N + 1;
    ShadeCtxLds<false> X; X.sob = (lds_u32 *)lds_sob; X.tris = nullptr; X.lights = nullptr; X.cfg = cfg; X.marg = marg;
    const uint32_t lane = __lane_id();
    for (uint32_t s = seg_next(ticket, Gn); s < Gn; s = seg_next(ticket, Gn)) {
        for (int m = 0; m < 6; ++m) {
            if (!(nee_kinds & (1u << m))) continue;
            const MatEntry *__restrict__ queue = Q.mat[m] + (size_t)s * seg_cap;
            const uint32_t n = rfl(*seg_count(Q, it, Q_MAT0 + m, Gn, s));
            for (uint32_t i = lane; i < n; i += 64u) {
                const MatEntry me = pslot(queue, i);
                const u4 stv = pslot(P.st, me.pid);
                const uint32_t field = f2u(pslot(P.ray_d[it & 1u], me.e).w) & ST_DIM_MASK;
                const VertexDims V = vertex_dims(field, true);
                const uint64_t index = (uint64_t)stv.x | ((uint64_t)stv.y << 32);
                const uint32_t dc[1] = {V.nee[4]};
                float uc[1];
                X.sobol<1>(S, index, dc, stv.w, uc);
                const float fl = floor_(uc[0] * (float)sc.n_lights); // the light choice of shade_item
                uint32_t li = fl > 0.0f ? (uint32_t)fl : 0u;
                if (li > sc.n_lights - 1u) li = sc.n_lights - 1u;
                if (li != env_li) continue;
                const uint32_t dn[2] = {V.nee[0], V.nee[1]};
                float u[2];
                X.sobol<2>(S, index, dn, stv.w, u);
                f3 wi, rgb; float pdf;
                const bool ok = inf_light_sample(sc, Le, mk2(u[0], u[1]), wi, pdf, rgb, &marg);
                pslot(P.pre0, me.e) = mkv4(wi, pdf);
                pslot(P.pre1, me.e) = mkv4(rgb, u2f(ok ? 1u : 0u));
            }
        }
    }
}

// totals[row*Q_STRIDE + q] = sum over segments of counts[(row*Q_STRIDE + q)*G + b]; one workgroup per (row, q)
__global__ __launch_bounds__(BLOCK) void k_reduce_counts(const uint32_t *__restrict__ counts, uint32_t G, uint32_t *__restrict__ totals) {
    __shared__ uint32_t acc;
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    uint32_t v = 0;
    for (uint32_t b = threadIdx.x; b < G; b += BLOCK) v += counts[(size_t)blockIdx.x * G + b];
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_down((int)v, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&acc, v);
    __syncthreads();
    if (threadIdx.x == 0) totals[blockIdx.x] = acc;
}

// Film gather, one workgroup per 16x16 tile of output pixels.  Per sample index the 20x20 neighbourhood of
// sample-pixels (p_film and radiance) is staged in LDS once and read by the 256 pixels of the tile, instead
// of every pixel fetching its 25 neighbours from HBM/L2 (16x less traffic; the first version of this kernel
// fetched 102 GB per frame).  Same sums, same order as film_item.
__global__ __launch_bounds__(BLOCK) void k_film(DParams R, DSampler S, DPaths P, const float *__restrict__ table, v4 *film, int32_t y0, int32_t y1, int32_t tiles_x) {
    __shared__ float tab[256];
    __shared__ v4 s_a[400]; __shared__ float s_lb[400]; // p_film.xy - 0.5, L.rg | L.b: one 16-byte and one 4-byte LDS read per candidate that contributes
    // The sample's footprint [p0x, p1x) x [p0y, p1y) (film.rs:66-73) against the tile, worked out ONCE per staged sample: bit c set when
    // the tile's column c is inside, bit 16 + r for row r.  A pixel's test is then `(mask & my two bits) == my two bits` -- two
    // instructions instead of film_weight's fourteen (four ceil / floor, four conversions, four comparisons ...) per (sample, pixel) pair.
    __shared__ uint32_t s_m[400];
    tab[threadIdx.x] = table[threadIdx.x];
    const int32_t tx0 = (int32_t)(blockIdx.x % (uint32_t)tiles_x) * 16, ty0 = y0 + (int32_t)(blockIdx.x / (uint32_t)tiles_x) * 16;
    const int32_t lx = (int32_t)(threadIdx.x & 15u), ly = (int32_t)(threadIdx.x >> 4);
    const int32_t x = tx0 + lx, y = ty0 + ly;
    const bool live = x < R.W && y < y1;
    const uint32_t my_bits = (1u << (uint32_t)lx) | (1u << (16u + (uint32_t)ly));
    v4 acc; acc.x = acc.y = acc.z = acc.w = 0.0f;
    if (live) acc = film[(size_t)y * (size_t)R.W + (size_t)x];
    const uint32_t npix = (uint32_t)(R.row1 - R.row0) * (uint32_t)R.NX;
    const uint32_t ns = R.s1 - R.s0;
    for (uint32_t k = 0; k < ns; ++k) {
        __syncthreads(); // previous round's reads are done (and `tab` is visible on the first round)
        for (uint32_t e = threadIdx.x; e < 400u; e += BLOCK) {
            const int32_t sx = tx0 - 2 + (int32_t)(e % 20u) - S.min_x, sy = ty0 - 2 + (int32_t)(e / 20u) - S.min_y;
            float pfx = -1.0e9f, pfy = -1.0e9f, lr = 0.0f, lg = 0.0f, lb = 0.0f; // far away: no pixel is in its footprint
            if (sx >= 0 && sx < R.NX && sy >= R.row0 && sy < R.row1) {
                const uint32_t pid = k * npix + (uint32_t)(sy - R.row0) * (uint32_t)R.NX + (uint32_t)sx;
                const f2a pf = pslot(P.pfilm, pid); const v4 Lv = pslot(P.L, pid);
                pfx = pf.x; pfy = pf.y; lr = Lv.x; lg = Lv.y; lb = Lv.z;
            }
            const float pdx = pfx - 0.5f, pdy = pfy - 0.5f; // film_weight's own first steps on the same values
            const int32_t p0x = (int32_t)ceil_(pdx - 2.0f), p0y = (int32_t)ceil_(pdy - 2.0f), p1x = (int32_t)(floor_(pdx + 2.0f) + 1.0f), p1y = (int32_t)(floor_(pdy + 2.0f) + 1.0f);
            auto span = [](int32_t lo, int32_t hi) { // bits [lo, hi) of a 16-bit field, both ends clipped to it
                const uint32_t l = (uint32_t)(lo < 0 ? 0 : (lo > 16 ? 16 : lo)), h = (uint32_t)(hi < 0 ? 0 : (hi > 16 ? 16 : hi));
                return h > l ? ((1u << h) - 1u) & ~((1u << l) - 1u) : 0u;
            };
            s_m[e] = span(p0x - tx0, p1x - tx0) | (span(p0y - ty0, p1y - ty0) << 16);
            v4 a; a.x = pdx; a.y = pdy; a.z = lr; a.w = lg; s_a[e] = a; s_lb[e] = lb;
        }
        __syncthreads();
        if (live) {
            for (int32_t dx = 0; dx < 5; ++dx)
                for (int32_t dy = 0; dy < 5; ++dy) {
                    const int32_t e = (ly + dy) * 20 + (lx + dx);
                    if ((s_m[e] & my_bits) != my_bits) continue; // x < p0x || x >= p1x || y < p0y || y >= p1y
                    const v4 a = s_a[e];
                    const float w = film_weight_inside(a.x, a.y, x, y, tab);
                    acc.x += a.z * w; acc.y += a.w * w; acc.z += s_lb[e] * w; acc.w += w;
                }
        }
    }
    if (live) film[(size_t)y * (size_t)R.W + (size_t)x] = acc;
}

// The stratified sampler's tables: one thread per 16x16 tile of the sample bounds, seeded with the tile's index like the
// reference's tile loop (integrator.rs:551-563: seed = tile.y * num_tiles.x + tile.x), sequential inside the tile.
__global__ __launch_bounds__(64) void k_strat_tables(int32_t NX, int32_t NY, uint32_t dim_ps, uint32_t n_dims, float *tab1, float *tab2) {
    const int32_t ntx = (NX + 15) / 16, nty = (NY + 15) / 16;
    const int32_t t = (int32_t)(blockIdx.x * 64u + threadIdx.x);
    if (t >= ntx * nty) return;
    const int32_t tx = t % ntx, ty = t / ntx;
    stratified_tile_tables((uint64_t)(ty * ntx + tx), tx * 16, min(tx * 16 + 16, NX), ty * 16, min(ty * 16 + 16, NY), NX, dim_ps, n_dims, tab1, tab2);
}

// The extension rays of round `it` as (o, d, +inf) records, whole 64-entry blocks of a segment together (the order a traversal wave meets them in)
__global__ __launch_bounds__(BLOCK) void k_dump_rays(DPaths P, DQueues Q, uint32_t it, uint32_t seg_cap, uint32_t G, uint32_t max_rays, float *out, uint32_t *n_out) {
    const uint32_t lane = threadIdx.x & 63u, s = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (s >= G) return;
    const uint32_t e0 = s * seg_cap, par = it & 1u;
    const uint32_t n = *seg_count(Q, it, Q_EXT, G, s);
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const uint32_t i = i0 + lane, cnt = n - i0 < 64u ? n - i0 : 64u;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(n_out, cnt);
        base = rfl(base);
        if (i < n && base + lane < max_rays) {
            const v4 o = pslot(P.ray_o[par], e0 + i), d = pslot(P.ray_d[par], e0 + i);
            float *r = out + (size_t)(base + lane) * 7u;
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = PT_INF;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_export_samples(DParams R, DSampler S, DPaths P, float *out) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t pid = blockIdx.x * BLOCK + threadIdx.x; pid < R.n_paths; pid += stride) {
        const PathCoord c = path_coord(R, S, pid);
        const size_t o = R.pixel_mode ? (size_t)c.s * 3 : (((size_t)c.sy * (size_t)R.NX + (size_t)c.sx) * S.spp + c.s) * 3;
        const v4 L = pslot(P.L, pid);
        out[o] = L.x; out[o + 1] = L.y; out[o + 2] = L.z;
    }
}

__global__ __launch_bounds__(BLOCK) void k_sobol(DSampler S, uint32_t n, const int32_t *px, const int32_t *py, const uint64_t *sn, const uint32_t *dims, float *out, uint64_t *idx_out) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t idx = sobol_index(S, sn[i], (uint32_t)(px[i] - S.min_x), (uint32_t)(py[i] - S.min_y));
    if (idx_out) idx_out[i] = idx;
    out[i] = sample_dimension(S, idx, dims[i], pixel_scramble(px[i], py[i]), px[i], py[i]);
}

// Self-test of the shared-divisor division (pt_vec.h, div_shared3) against the compiler's IEEE division, bit for bit, on
// operand sets drawn from a counter-based generator.  mode 0: raw random bit patterns (every class at its natural share: 1/256 each of
// zeros + denormals and infinities + NaNs); 1: exponents around every threshold of the fast path's window and of v_div_scale /
// v_div_fixup, mantissas random / all zeros / all ones, signed zeros; 2: the ranges of a render (pdf-like divisors 2^-27 .. 2^13,
// radiance-like numerators, a quarter of them zero); 3: beta / (1 - q) of the Russian roulette; 4: quotients next to 1 and next to
// rounding ties (a = b x small factors, a = b +- ulps).
__device__ inline uint64_t mix64(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
__device__ inline uint32_t edge_bits(uint32_t r) { // r: 32 random bits
    const uint32_t exps[32] = {0, 0, 1, 2, 22, 23, 24, 25, 31, 32, 33, 78, 79, 80, 81, 82, 126, 127, 128, 150, 172, 173, 174, 175, 176, 221, 222, 223, 252, 253, 254, 255};
    const uint32_t e = exps[r & 31u], how = (r >> 5) & 3u, sign = (r >> 7) & 1u;
    const uint32_t man = how == 0u ? 0u : (how == 1u ? 0x7fffffu : ((r >> 9) & 0x7fffffu));
    return (sign << 31) | (e << 23) | man;
}
__global__ __launch_bounds__(BLOCK) void k_selftest_div3(uint64_t seed, uint32_t mode, uint64_t per_thread, unsigned long long *out /* [0] mismatches, [1] fast-path sets */, uint32_t *first_bad /* 10 words + a lock */) {
    const uint64_t tid = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long bad = 0, fast = 0;
    for (uint64_t k = 0; k < per_thread; ++k) {
        const uint64_t h0 = mix64(seed ^ (tid * per_thread + k) * 0xd1342543de82ef95ull), h1 = mix64(h0);
        uint32_t w[4] = {(uint32_t)h0, (uint32_t)(h0 >> 32), (uint32_t)h1, (uint32_t)(h1 >> 32)};
        if (mode == 1u) { for (int j = 0; j < 4; ++j) w[j] = edge_bits(w[j]); }
        else if (mode == 2u) {
            w[3] = ((100u + (w[3] >> 24) % 41u) << 23) | (w[3] & 0x7fffffu);
            for (int j = 0; j < 3; ++j) w[j] = (w[j] >> 30) == 0u ? (w[j] & 0x80000000u) : (((90u + (w[j] >> 24) % 51u) << 23) | (w[j] & 0x7fffffu) | ((w[j] >> 29) == 7u ? 0x80000000u : 0u));
        } else if (mode == 3u) {
            const float mx = (float)(w[3] >> 8) * 0x1p-24f;
            const float q = max_nz(0.05f, 1.0f - mx);
            w[3] = f2u(1.0f - q);
            for (int j = 0; j < 3; ++j) w[j] = f2u((float)(w[j] >> 8) * 0x1p-23f * ((w[j] & 255u) == 0u ? 0.0f : 1.0f));
        } else if (mode == 4u) {
            const uint32_t b = ((80u + (w[3] >> 24) % 90u) << 23) | (w[3] & 0x7fffffu) | ((w[3] >> 23 & 1u) << 31);
            w[3] = b;
            for (int j = 0; j < 3; ++j) {
                const uint32_t sel = w[j] & 7u, d = (w[j] >> 3) & 15u;
                const float fb = u2f(b);
                float a = fb;
                if (sel == 0u) a = u2f(b + d); else if (sel == 1u) a = u2f(b - d); else if (sel == 2u) a = fb * (float)(1u + d); else if (sel == 3u) a = fb * (1.0f + (float)d * 0x1p-23f);
                else if (sel == 4u) a = fb * 0x1.fffffep-1f; else if (sel == 5u) a = fb * 0.5f + u2f((f2u(fb) & 0xff800000u) - (12u << 23)); else if (sel == 6u) a = u2f((b & 0xff800000u) | 0x7fffffu); else a = u2f(b & 0xff800000u);
                w[j] = f2u(a);
            }
        }
        const float ax = u2f(w[0]), ay = u2f(w[1]), az = u2f(w[2]), b = u2f(w[3]);
        const f3 got = div_shared3(mk3(ax, ay, az), b);
        const float wx = ax / b, wy = ay / b, wz = az / b; // the compiler's IEEE division
        { // (how many sets took the fast path: the test wants to know that it was exercised)
            const uint32_t LO = 80u << 23, HI = 175u << 23, ux = w[0] & 0x7fffffffu, uy = w[1] & 0x7fffffffu, uz = w[2] & 0x7fffffffu, us = w[3] & 0x7fffffffu;
            auto okn = [&](uint32_t u) { return u == 0u || (u >= LO && u < HI); };
            if (us >= LO && us < HI && okn(ux) && okn(uy) && okn(uz)) ++fast;
        }
        if (f2u(got.x) != f2u(wx) || f2u(got.y) != f2u(wy) || f2u(got.z) != f2u(wz)) {
            ++bad;
            if (atomicCAS(first_bad + 10, 0u, 1u) == 0u) { first_bad[0] = w[0]; first_bad[1] = w[1]; first_bad[2] = w[2]; first_bad[3] = w[3]; first_bad[4] = f2u(got.x); first_bad[5] = f2u(got.y); first_bad[6] = f2u(got.z); first_bad[7] = f2u(wx); first_bad[8] = f2u(wy); first_bad[9] = f2u(wz); }
        }
    }
    if (bad) atomicAdd(out, bad);
    atomicAdd(out + 1, fast);
}

// ---- device buffers ---------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return PTRS_OK;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        HIPCHK(hipMalloc(&p, n));
        bytes = n;
        return PTRS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

template <class T> int upload(DevBuf &b, const std::vector<T> &v) {
    size_t n = v.size() * sizeof(T);
    int rc = b.ensure(n ? n : 16);
    if (rc != PTRS_OK) return rc;
    if (n) HIPCHK(hipMemcpy(b.p, v.data(), n, hipMemcpyHostToDevice));
    return PTRS_OK;
}

struct SobolDevice { // one copy per device
    int device = -1;
    DevBuf matrices, vdc, vdc_inv, bytetab, nibtab;
    uint32_t stride = 52;
};
std::mutex g_mu;
std::vector<SobolDevice *> g_sobol;

int get_sobol(int device, SobolDevice **out) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto *s : g_sobol) if (s->device == device) { *out = s; return PTRS_OK; }
    const unsigned char *b = k_sobol_blob;
    if (sizeof(k_sobol_blob) < 32 || std::memcmp(b, "PTRSSOB1", 8) != 0) { g_err = "embedded sobol tables corrupt"; return PTRS_ERR_INVALID; }
    uint32_t hdr[6]; std::memcpy(hdr, b + 8, 24);
    const size_t nmat = (size_t)hdr[0] * hdr[1], stride = hdr[4];
    const unsigned char *pm = b + 32, *pv = pm + nmat * 4 + 52 * 4, *pvi = pv + (size_t)hdr[2] * stride * 8;
    auto *s = new SobolDevice(); s->device = device; s->stride = (uint32_t)stride;
    int rc;
    const size_t mat_pad = 16 * 52 * 4; // a path that overruns the 1024 dimensions reads up to 8 rows past the table before its kernel raises PTRS_ERRFLAG_SOBOL_DIM
    if ((rc = s->matrices.ensure(nmat * 4 + mat_pad)) != PTRS_OK || (rc = s->vdc.ensure((size_t)hdr[2] * stride * 8)) != PTRS_OK || (rc = s->vdc_inv.ensure((size_t)hdr[3] * stride * 8)) != PTRS_OK) { delete s; return rc; }
    HIPCHK(hipMemset(s->matrices.p, 0, nmat * 4 + mat_pad));
    HIPCHK(hipMemcpy(s->matrices.p, pm, nmat * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->vdc.p, pv, (size_t)hdr[2] * stride * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->vdc_inv.p, pvi, (size_t)hdr[3] * stride * 8, hipMemcpyHostToDevice));
    {
        SobolTablesHost T; T.matrices.resize(nmat); std::memcpy(T.matrices.data(), pm, nmat * 4);
        std::vector<uint32_t> bt; build_sobol_bytetab(T, bt);
        if ((rc = s->bytetab.ensure(bt.size() * 4)) != PTRS_OK) { delete s; return rc; }
        HIPCHK(hipMemcpy(s->bytetab.p, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
        // nibtab[dim][n][v] = XOR of the matrix columns 4n + j over the set bits j of v: the form the shade kernels stage into
        // LDS (64 B per index nibble and dimension; the byte tables above would need 1 KB)
        std::vector<uint32_t> nt((size_t)1024 * SOBOL_NIBBLES * 16, 0u);
        for (uint32_t d = 0; d < 1024; ++d)
            for (uint32_t n = 0; n < (uint32_t)SOBOL_NIBBLES; ++n)
                for (uint32_t v = 0; v < 16; ++v) {
                    uint32_t x = 0;
                    for (uint32_t j = 0; j < 4; ++j) if ((v >> j) & 1u) { const uint32_t col = 4 * n + j; if (col < 52) x ^= T.matrices[(size_t)d * 52 + col]; }
                    nt[((size_t)d * SOBOL_NIBBLES + n) * 16 + v] = x;
                }
        if ((rc = s->nibtab.ensure(nt.size() * 4)) != PTRS_OK) { delete s; return rc; }
        HIPCHK(hipMemcpy(s->nibtab.p, nt.data(), nt.size() * 4, hipMemcpyHostToDevice));
    }
    g_sobol.push_back(s);
    *out = s;
    return PTRS_OK;
}

} // namespace

enum { MAX_LANES = 8 };
struct PtrsScene {
    int device = 0;
    HostScene H; // host copy kept for validation / stats
    DScene sc{};
    DevBuf nodes2, nodes4, nodes, tris, shade, mats, texs, levels, texdata, lights, distdata, inf;
    DevBuf stack_spill;  // global part of the traversal stacks (trees deeper than the LDS column), one column per resident thread
    StackSpill spill{nullptr, 0};
    size_t spill_lane_elems = 0;
    std::map<std::string, int> overrides;  // ptrs_scene_set_option: this scene's renders take these instead of the process-wide values
    std::map<const void *, int> occupancy; // workgroups per CU by kernel (hipOccupancyMaxActiveBlocksPerMultiprocessor), asked once
    uint32_t stack_lds = 16; // LDS stack entries per lane: 8 when the tree allows it, else 16 (+ spill)
    std::vector<double> alive_frac; // survival profile of this scene's paths: share of a pass's paths that reached round k, from the last finished pass (HipBackend::learn); empty until a pass has finished
    // render workspace, grown on demand and reused across calls
    DevBuf ws[MAX_LANES][32];   // per pipeline lane
    DevBuf counts[MAX_LANES], totals[MAX_LANES], tickets[MAX_LANES];
    DevBuf stats, table, film_tmp, samples_tmp, strat1, strat2, row_cost;
    hipStream_t lane_stream[MAX_LANES] = {}; // lane 0 runs on the caller's stream, the others on these
    hipEvent_t lane_ev[MAX_LANES] = {};      // film-done per lane
    std::vector<hipEvent_t> ev_pool;
    int n_cu = 256;
    ~PtrsScene() {
        for (auto &b : {&stack_spill, &nodes2, &nodes4, &nodes, &tris, &shade, &mats, &texs, &levels, &texdata, &lights, &distdata, &inf, &stats, &table, &film_tmp, &samples_tmp, &strat1, &strat2, &row_cost}) b->release();
        for (auto &b : counts) b.release();
        for (auto &b : totals) b.release();
        for (auto &b : tickets) b.release();
        for (auto &l : ws) for (auto &b : l) b.release();
        for (auto st : lane_stream) if (st) (void)hipStreamDestroy(st);
        for (auto e : lane_ev) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_pool) (void)hipEventDestroy(e);
    }
};

namespace {

struct HipBackend {
    PtrsScene *ps; hipStream_t stream; SobolDevice *sob;
    DScene sc; DSampler S; DCamera C; DParams R; DPaths P; DQueues Q;
    uint32_t cap = 0, rows = 0, depth = 0, flags = 0, kinds_mask = 0;
    int feat = FEAT_FULL, feat_trace = FEAT_FULL;
    uint32_t geom4 = 0xffffffffu; // 16-byte vectors needed to hold nodes + triangles in LDS
    uint32_t G = 1, seg_cap = 0;  // segmented queues of the current pass
    // pipeline lanes: the members above (stream, R, P, Q, G, seg_cap) are those of the selected lane
    struct Lane { hipStream_t stream; DParams R; DPaths P; DQueues Q; uint32_t G, seg_cap; };
    Lane lane_[MAX_LANES]; uint32_t n_lanes = 1, cur = 0;
    hipEvent_t film_prev = nullptr;
    Options opt;
    // Pipeline lanes of this render and the size of its launches (called before begin()).  Defaults (lanes = grid_mult = grid_pct = 0):
    // FOUR lanes whose passes have 2 048 segments each (grid_mult 1): a queue kernel then holds 2 048 waves, a third of the machine's
    // slots, for its whole run, and the kernels of the four lanes -- traversal and shade kernels of different passes, with their
    // different appetites for registers, LDS and memory -- fill the CUs together all the time.  Measured against this round's other
    // arrangements on one box: Cornell 161.2-161.6 ms (three lanes with 16 384 segments at half the resident capacity each: 169.2;
    // one lane: 192), colonnade 114.9 (one lane with 16 384 segments: 116.9), classroom 847-852 (852); five and more lanes are
    // slower (182 / 174 / 167 ms for 5 / 6 / 8 on Cornell: the runtime maps streams onto four hardware queues).  A job too small to
    // give four lanes a pass each runs on one lane with 16 384 segments, and so does every single-lane render (the profiled frames).
    uint32_t lanes(uint64_t job_paths = 0, const bool * = nullptr, bool single_pixel = false, uint32_t spp = 0xffffffffu) {
        int want = opt.lanes;
        if (want == 0) want = (single_pixel || spp < 2u) ? 1 : (job_paths >= (4ull << 20) ? 4 : 2); // (a small job of ONE sample per pixel is one pass: one lane with its 16 384 segments, where the fused tail takes the whole pass -- the band planner's probe)
        // (render_single_pixel traces spp paths: one lane, the whole workspace; a job under 4 M paths: two lanes -- with the fused tail a small pass is a handful of launches, and two of them side by side beat one: 1 / 4 rows of the Cornell frame 3.5 / 4.3 -> 2.8 / 3.4 ms)
        if (share > 1 && opt.lanes == 0) want = std::max(1, want / share);        // renders that share a device (ptrs_render_multi replicas) share its hardware queues too
        n_lanes = (uint32_t)(want < 1 ? 1 : (want > MAX_LANES ? MAX_LANES : want));
        return n_lanes;
    }
    // Segments per pass, once the passes are planned (begin(): `max_paths` is the size of the largest pass, whatever auto_capacity,
    // paths_per_pass or the row-chunk plan made of the job).  Defaults: CUs x 8 with several lanes, CUs x 64 with one.
    // Scenes whose tree lives in HBM / L2 (quad form) wait for node fetches and gain from more waves of a kernel in flight once a
    // segment still holds ~10 k paths: classroom (66 M paths per pass) 801 -> 797 / 784 / 795 ms at 4 096 / 6 144 / 8 192 segments,
    // colonnade (15 M per pass) 107.3 -> 110.2 / 112.9 / 115.9: it stays at 2 048.  (The LDS-resident Cornell, bound by what the
    // SIMDs issue, loses 1 % at 4 096.)
    void plan_segments(uint64_t max_paths) {
        if (opt.grid_mult == 0) {
            opt.grid_mult = n_lanes > 1 ? 1 : 8;
            if (n_lanes > 1 && ps->sc.n_nodes4 != 0) {
                const uint64_t per_unit = (uint64_t)ps->n_cu * 8u * 10240u;
                const uint64_t k = (max_paths + per_unit / 2) / per_unit;
                opt.grid_mult = (int)(k < 1 ? 1 : (k > 4 ? 4 : k));
            }
            if (opt.grid_pct == 0) opt.grid_pct = 100;
        }
        if (opt.grid_pct == 0) opt.grid_pct = n_lanes > 1 ? 50 : 100; // (several lanes with many segments each: a launch takes half of its resident capacity)
    }
    void select(uint32_t l) {
        if (l == cur) return;
        lane_[cur] = Lane{stream, R, P, Q, G, seg_cap};
        cur = l; stream = lane_[l].stream; R = lane_[l].R; P = lane_[l].P; Q = lane_[l].Q; G = lane_[l].G; seg_cap = lane_[l].seg_cap;
    }
    // paths per pass and lane that keep the whole workspace (17-19 state vectors + the queues per path, all lanes) inside
    // opt.workspace_pct of the memory that is free now plus what this scene already holds from earlier renders
    uint64_t auto_capacity(uint32_t lanes_n, const bool *kinds) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 1ull << 27;
        size_t held = 0;
        for (auto &l : ps->ws) for (auto &b : l) held += b.bytes;
        uint32_t nk = 0; for (int k = 0; k < 7; ++k) nk += kinds[k] ? 1u : 0u;
        const double per_path = (17.0 + (ps->H.inf_lights.empty() ? 0.0 : 2.0)) * 16.0 + 8.0 + 4.0 + 3.0 * 4.0 + nk * 8.0; // 15 vectors in queue order (ray, throughput double-buffered), 2 by path slot, the presampled light sample, the film position, pre_z, three queues, the shade queues
        const double budget = (double)(fr + held) * (double)opt.workspace_pct / 100.0 / (double)std::max(1, share); // `share` renders divide this device's memory (ptrs_render_multi)
        const double cap = budget / (per_path * (double)lanes_n);
        return cap < 65536.0 ? 65536ull : (uint64_t)cap;
    }
    int share = 1;
    int grid_max = 8192;
    uint32_t refill_connect = 16;
    bool vote = true, vote_connect = true;
    uint32_t refill = 16; // idle-lane threshold of the lane-refill kernels (64: a wave takes new rays only when all its lanes are idle)
    int rc = PTRS_OK;
    // timing
    struct Span { int cat; hipEvent_t a, b; };
    std::vector<Span> spans; size_t ev_next = 0;
    uint64_t launches = 0, trace_launches = 0;

    const uint32_t *sobol_matrices() { return (const uint32_t *)sob->matrices.p; }
    const uint32_t *sobol_bytetab() { return (const uint32_t *)sob->bytetab.p; }
    const uint32_t *sobol_nibtab() { return (const uint32_t *)sob->nibtab.p; }
    int strat_tables(int32_t NX, int32_t NY, uint32_t dim_ps, uint32_t n_dims, const float **t1, const float **t2, std::string &err) {
        const size_t n = (size_t)NX * (size_t)NY * n_dims * dim_ps * dim_ps;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess && n * 12 > (fr + ps->strat1.bytes + ps->strat2.bytes) / 2) { err = "stratified sampler: the per-pixel tables (" + std::to_string(n * 12 >> 20) + " MiB) do not fit"; return PTRS_ERR_UNSUPPORTED; }
        int rc;
        if ((rc = ps->strat1.ensure(n * 4)) != PTRS_OK || (rc = ps->strat2.ensure(n * 8)) != PTRS_OK) { err = g_err; return rc; }
        const int32_t tiles = ((NX + 15) / 16) * ((NY + 15) / 16);
        hipLaunchKernelGGL(k_strat_tables, dim3((uint32_t)(tiles + 63) / 64), dim3(64), 0, stream, NX, NY, dim_ps, n_dims, (float *)ps->strat1.p, (float *)ps->strat2.p);
        if (hipStreamSynchronize(stream) != hipSuccess || hipGetLastError() != hipSuccess) { err = "stratified table kernel failed"; return PTRS_ERR_DEVICE; }
        *t1 = (const float *)ps->strat1.p; *t2 = (const float *)ps->strat2.p;
        return PTRS_OK;
    }
    const uint64_t *sobol_vdc(uint32_t row) { return (const uint64_t *)sob->vdc.p + (size_t)row * sob->stride; }
    const uint64_t *sobol_vdc_inv(uint32_t row) { return (const uint64_t *)sob->vdc_inv.p + (size_t)row * sob->stride; }

    hipEvent_t ev() {
        if (ev_next == ps->ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { rc = PTRS_ERR_DEVICE; return nullptr; } ps->ev_pool.push_back(e); }
        return ps->ev_pool[ev_next++];
    }
    enum { T_EXTEND = 0, T_AUX = 1, T_FILM = 2, T_CONNECT = 3, T_SHADE = 4, T_TAIL = 5, T_NUM = 6 }; // kernel classes of the timing spans
    uint64_t cat_launches[T_NUM] = {0, 0, 0, 0, 0, 0};
    void t0(int cat) { if (flags & PTRS_FLAG_TIMING) { Span s; s.cat = cat; s.a = ev(); s.b = ev(); if (s.a) (void)hipEventRecord(s.a, stream); spans.push_back(s); } ++launches; ++cat_launches[cat]; if (cat == T_EXTEND || cat == T_CONNECT) ++trace_launches; }
    void t1() { if (flags & PTRS_FLAG_TIMING) { if (spans.back().b) (void)hipEventRecord(spans.back().b, stream); } }
    int grid_for(uint32_t n) const { uint32_t g = (n + BLOCK - 1) / BLOCK; if (g < 1) g = 1; const uint32_t gm = (uint32_t)ps->n_cu * 8u; return (int)(g > gm ? gm : g); }

    // Workgroups of a persistent queue kernel: what fits the machine at once (the kernel's resident workgroups per CU x CUs), never
    // more than the segments need (4 waves = 4 segments per workgroup).  A launch that over-estimates the residency loses nothing:
    // workgroups that start late find the ticket counter exhausted and leave.
    uint32_t last_grid[T_NUM] = {0, 0, 0, 0, 0, 0}, last_per_cu[T_NUM] = {0, 0, 0, 0, 0, 0}; // last launch of each class (PtrsStats)
    template <class F> int wgs_per_cu(F fn) {
        const void *key = reinterpret_cast<const void *>(fn);
        auto itr = ps->occupancy.find(key);
        if (itr != ps->occupancy.end()) return opt.persist ? itr->second : 8;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
        ps->occupancy[key] = nb > 8 ? 8 : nb;
        return opt.persist ? ps->occupancy[key] : 8;
    }
    template <class F> uint32_t resident_grid(F fn) { return std::max(1u, (uint32_t)ps->n_cu * (uint32_t)wgs_per_cu(fn) * (uint32_t)(opt.grid_pct ? opt.grid_pct : 100) / 100u); } // workgroups a launch of fn holds
    template <class F> uint32_t persistent_grid(F fn, int cat) {
        const uint32_t need = (G + WAVES - 1) / WAVES, fit = resident_grid(fn);
        last_grid[cat] = need < fit ? need : fit; last_per_cu[cat] = (uint32_t)wgs_per_cu(fn);
        return last_grid[cat];
    }
    // Segments per pass: about CUs x 8 x grid_mult, but a WHOLE multiple of the waves the traversal kernels hold (and of the first shade
    // kernel's, where one number serves both).  A launch of W resident waves works through G equal segments in ceil(G / W) rounds, the last
    // one only partly filled: at G / W = 2.67 a third of the waves idle for the last third of every kernel (utilisation G / W / ceil(G / W) =
    // 0.89; the same at 5.33), at a whole ratio none do.
    template <int FEAT_T> uint32_t whole_rounds(uint32_t target) {
        const uint32_t we = resident_grid(pick_extend<FEAT_T>(vote, ps->spill.p != nullptr)) * WAVES;
        uint32_t ws = 0;
        for (int k = 0; k < 6 && !ws; ++k) if (ps->H.kinds_present[k]) ws = resident_grid(shade_fn(k)) * WAVES;
        auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
        uint64_t unit = we;
        if (ws) { const uint64_t l = (uint64_t)we / gcd(we, ws) * ws; if (l <= 2ull * target) unit = l; }
        const uint64_t k = std::max<uint64_t>(1, (target + unit / 2) / unit);
        return (uint32_t)std::min<uint64_t>(k * unit, 4ull * target);
    }
    uint32_t *ticket(uint32_t it, int which) { return Q.tickets + ((size_t)it * Q_STRIDE + (size_t)which) * TK_LAUNCH_WORDS; }

    int begin(const DScene &sc_, const DSampler &S_, const DCamera &C_, uint32_t capacity, uint32_t count_rows, uint32_t bvh_depth, uint32_t flags_, int feat_, int feat_trace_, std::string &err) {
        sc = sc_; S = S_; C = C_; cap = capacity; rows = count_rows; depth = bvh_depth; flags = flags_; feat = feat_; feat_trace = feat_trace_;
        plan_segments(capacity);
        // phase voting: quad-node scenes gain in both traversal kernels; on the LDS pair form a step is cheap enough that the vote's
        // own instructions eat the gain in the connect kernel (+20 %), the extension kernel keeps 4 % (A/B on MI355X)
        vote = opt.vote >= 0 ? opt.vote != 0 : true;
        vote_connect = opt.vote >= 0 ? opt.vote == 1 : sc.n_nodes4 != 0;
        // idle-lane threshold: a voting wave comes back for retire / refill in batches, and with the cheap steps of the kernels without
        // alpha masks a bigger batch pays (Cornell extend 88.8 -> 85.7 ms, colonnade connect 38.2 -> 36.4 ms at 32); the full-feature
        // kernels (classroom) are better off at 16 (extend 227 vs 233 ms).  0 = no refill while a lane still works (threshold 64).
        refill = (uint32_t)(opt.refill > 0 ? opt.refill : (opt.refill == 0 ? 64 : ((vote && feat_trace == FEAT_SIMPLE) ? 32 : 16)));
        refill_connect = (uint32_t)(opt.refill_connect > 0 ? opt.refill_connect : (opt.refill_connect == 0 ? 64 : ((vote_connect && feat_trace == FEAT_SIMPLE) ? 32 : 16)));
        geom4 = sc.n_nodes4 ? 0xffffffffu : LN_V4 * sc.n_nodes2 + 9u * sc.n_prims; // quad form: global kernels; pair form: fits the LDS staging area by construction (pt_host_scene.h)
        for (int k = 0; k < 7; ++k) if (ps->H.kinds_present[k]) kinds_mask |= 1u << k;
        grid_max = ps->n_cu * 8 * (opt.grid_mult ? opt.grid_mult : 8);
        if (opt.persist && opt.whole_rounds) grid_max = (int)(feat_trace == FEAT_FULL ? whole_rounds<FEAT_FULL>((uint32_t)grid_max) : (feat_trace == FEAT_IMG_ENV ? whole_rounds<FEAT_IMG_ENV>((uint32_t)grid_max) : whole_rounds<FEAT_SIMPLE>((uint32_t)grid_max)));
        const size_t n16 = (size_t)cap * 16, n4 = ((size_t)cap + ((size_t)grid_max + 1) * 64) * 4; // queues: G segments of whole 64-entry chunks
        if ((rc = ps->stats.ensure(CNT_NUM * 8)) != PTRS_OK || (rc = ps->table.ensure(1024)) != PTRS_OK) { err = g_err; return rc; }
        if (ps->spill_lane_elems) { // the traversal stacks' global columns: one set per pipeline lane of THIS render (concurrent lanes must not share columns)
            if ((rc = ps->stack_spill.ensure(ps->spill_lane_elems * n_lanes * sizeof(unsigned long long))) != PTRS_OK) { err = g_err + " (traversal stack spill columns)"; return rc; }
            ps->spill.p = (unsigned long long *)ps->stack_spill.p;
        }
        for (uint32_t l = 0; l < n_lanes && n_lanes > 1; ++l) {
            if (l > 0 && !ps->lane_stream[l] && hipStreamCreateWithFlags(&ps->lane_stream[l], hipStreamNonBlocking) != hipSuccess) { err = "cannot create a pipeline stream"; return PTRS_ERR_DEVICE; }
            if (!ps->lane_ev[l] && hipEventCreateWithFlags(&ps->lane_ev[l], hipEventDisableTiming) != hipSuccess) { err = "cannot create pipeline events"; return PTRS_ERR_DEVICE; }
        }
        const size_t nq = n4 / 4; // positions of a queue, and of every array in queue order
        const bool presampled = presample_li() != 0xffffffffu;
        for (uint32_t l = 0; l < n_lanes; ++l) {
            DPaths Pl; DQueues Ql;
            std::memset(&Pl, 0, sizeof(Pl));
            int w = 0;
            auto take = [&](size_t bytes, void **out, const char *what) -> int {
                int r = ps->ws[l][w].ensure(bytes);
                if (r != PTRS_OK) { err = g_err + " (" + what + " of pipeline lane " + std::to_string(l) + "; ptrs_set_option(\"lanes\", 1) or a smaller workspace_pct shrink the workspace)"; return r; }
                *out = ps->ws[l][w++].p;
                return PTRS_OK;
            };
            void **by_queue16[] = {(void **)&Pl.ray_o[0], (void **)&Pl.ray_o[1], (void **)&Pl.ray_d[0], (void **)&Pl.ray_d[1], (void **)&Pl.beta[0], (void **)&Pl.beta[1], (void **)&Pl.hit,
                                   (void **)&Pl.nee0, (void **)&Pl.nee1, (void **)&Pl.nee2, (void **)&Pl.sh_o, (void **)&Pl.sh_d, (void **)&Pl.mis_o, (void **)&Pl.mis_d, (void **)&Pl.nhit};
            for (auto sl : by_queue16) if ((rc = take(nq * 16, sl, "path state")) != PTRS_OK) return rc;
            void **by_slot16[] = {(void **)&Pl.L, (void **)&Pl.st};
            for (auto sl : by_slot16) if ((rc = take(n16, sl, "path state")) != PTRS_OK) return rc;
            if ((rc = take((size_t)cap * 8, (void **)&Pl.pfilm, "path state")) != PTRS_OK) return rc;
            if ((rc = take(nq * 4, (void **)&Pl.pre_z, "path state")) != PTRS_OK) return rc;
            if (presampled) { if ((rc = take(nq * 16, (void **)&Pl.pre0, "path state")) != PTRS_OK || (rc = take(nq * 16, (void **)&Pl.pre1, "path state")) != PTRS_OK) return rc; } else w += 2;
            void **slots4[] = {(void **)&Ql.ext[0], (void **)&Ql.ext[1], (void **)&Ql.nee};
            for (auto sl : slots4) if ((rc = take(n4, sl, "queues")) != PTRS_OK) return rc;
            for (int k = 0; k < Q_NUM_MAT; ++k) {
                Ql.mat[k] = nullptr;
                if (kinds_mask & (1u << k)) { if ((rc = take(nq * sizeof(MatEntry), (void **)&Ql.mat[k], "queues")) != PTRS_OK) return rc; } else ++w;
            }
            if ((rc = ps->counts[l].ensure((size_t)rows * Q_STRIDE * (size_t)grid_max * 4)) != PTRS_OK || (rc = ps->totals[l].ensure((size_t)rows * Q_STRIDE * 4)) != PTRS_OK ||
                (rc = ps->tickets[l].ensure(((size_t)rows * Q_STRIDE * TK_LAUNCH_WORDS + rows) * 4)) != PTRS_OK) { err = g_err; return rc; }
            Ql.counts = (uint32_t *)ps->counts[l].p; Ql.stats = (unsigned long long *)ps->stats.p; Ql.tickets = (uint32_t *)ps->tickets[l].p; Ql.alive = Ql.tickets + (size_t)rows * Q_STRIDE * TK_LAUNCH_WORDS;
            lane_[l] = Lane{l == 0 ? stream : ps->lane_stream[l], DParams{}, Pl, Ql, 1u, 0u};
        }
        cur = 0; P = lane_[0].P; Q = lane_[0].Q;
        float tab[256]; gaussian_filter_table(tab);
        if (hipMemcpyAsync(ps->table.p, tab, 1024, hipMemcpyHostToDevice, stream) != hipSuccess || hipMemsetAsync(Q.stats, 0, CNT_NUM * 8, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { err = "workspace initialisation failed"; return PTRS_ERR_DEVICE; }
        // fork: the second lane starts after everything the caller had queued on its stream (now drained) -- nothing to wait for
        film_prev = nullptr;
        return PTRS_OK;
    }
    // per pass: G queue segments (one wave each), each holds at most seg_cap entries
    void pass_begin(const DParams &R_) {
        R = R_;
        const uint32_t chunks = (R.n_paths + 63u) / 64u;
        G = chunks < (uint32_t)grid_max ? chunks : (uint32_t)grid_max;
        if (G < 1) G = 1;
        seg_cap = ((chunks + G - 1) / G) * 64u;
        (void)hipMemsetAsync(Q.counts, 0, (size_t)rows * Q_STRIDE * G * 4, stream);
        (void)hipMemsetAsync(Q.tickets, 0, ((size_t)rows * Q_STRIDE * TK_LAUNCH_WORDS + rows) * 4, stream); // (+ the alive flags behind them)
    }
    void generate() { t0(T_AUX); hipLaunchKernelGGL(k_generate, dim3((G + WAVES - 1) / WAVES), dim3(BLOCK), 0, stream, R, S, C, P, Q, seg_cap, G, (uint32_t)deal_by_region()); t1(); }
    int deal_by_region() const { return opt.deal; } // (option `deal`: measured slower, off)

    // the instantiation of a traversal kernel for this scene: LDS stack depth, spill columns, geometry source, phase voting
    typedef void (*TravFn)(DParams, DScene, StackSpill, DPaths, DQueues, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t *);
#define PTRS_PICK(K, D, O, GE) (v ? (TravFn)K<FEAT, D, O, GE, true> : (TravFn)K<FEAT, D, O, GE, false>)
#define PTRS_PICK_ALL(K) (ps->stack_lds == 9 ? PTRS_PICK(K, 9, false, 544) : ps->stack_lds == 8 ? (geom4 <= 640 ? (ovf ? PTRS_PICK(K, 8, true, 640) : PTRS_PICK(K, 8, false, 640)) : geom4 <= 1536 ? (ovf ? PTRS_PICK(K, 8, true, 1536) : PTRS_PICK(K, 8, false, 1536)) : (ovf ? PTRS_PICK(K, 8, true, 0) : PTRS_PICK(K, 8, false, 0))) \
                                              : (ovf ? PTRS_PICK(K, 16, true, 0) : PTRS_PICK(K, 16, false, 0)))
    template <int FEAT> TravFn pick_extend(bool v, bool ovf) { return PTRS_PICK_ALL(k_extend_rf); }
    template <int FEAT> TravFn pick_connect(bool v, bool ovf) { return PTRS_PICK_ALL(k_connect_rf); }
#undef PTRS_PICK_ALL
#undef PTRS_PICK
    StackSpill lane_spill() { StackSpill sp = ps->spill; if (sp.p) sp.p += (size_t)cur * ps->spill_lane_elems; return sp; } // this lane's columns

    template <int FEAT> void extend_t(uint32_t it) {
        const StackSpill sp = lane_spill();
        const uint32_t epi_mask = opt.fused_epilogue ? kinds_mask : 0u; // non-zero: the kernel runs its segment's epilogue behind its last ray
        const TravFn fn = pick_extend<FEAT>(vote, sp.p != nullptr);
        hipLaunchKernelGGL(fn, dim3(persistent_grid(fn, T_EXTEND)), dim3(BLOCK), 0, stream, R, sc, sp, P, Q, it, seg_cap, refill, epi_mask, G, ticket(it, TK_EXTEND));
        if (epi_mask) return;
        t1(); t0(T_AUX); // the traversal span ends here: the epilogue is shading-side work
        hipLaunchKernelGGL((k_epilogue<FEAT>), dim3(persistent_grid(k_epilogue<FEAT>, T_AUX)), dim3(BLOCK), 0, stream, R, sc, P, Q, it, kinds_mask, seg_cap, G, ticket(it, TK_EPILOGUE));
    }
    void extend(uint32_t it) { t0(T_EXTEND); if (feat_trace == FEAT_FULL) extend_t<FEAT_FULL>(it); else if (feat_trace == FEAT_IMG_ENV) extend_t<FEAT_IMG_ENV>(it); else extend_t<FEAT_SIMPLE>(it); t1(); }
    template <int FEAT> void connect_t(uint32_t it) {
        const StackSpill sp = lane_spill();
        const uint32_t fused = (opt.fused_epilogue && opt.fused_resolve) ? 1u : 0u;
        const TravFn fn = pick_connect<FEAT>(vote_connect, sp.p != nullptr);
        hipLaunchKernelGGL(fn, dim3(persistent_grid(fn, T_CONNECT)), dim3(BLOCK), 0, stream, R, sc, sp, P, Q, it, seg_cap, refill_connect, fused, G, ticket(it, TK_CONNECT));
        if (fused) return;
        t1(); t0(T_AUX);
        hipLaunchKernelGGL((k_resolve<FEAT>), dim3(persistent_grid(k_resolve<FEAT>, T_AUX)), dim3(BLOCK), 0, stream, sc, P, Q, it, seg_cap, G, ticket(it, TK_RESOLVE));
    }
    void connect(uint32_t it) { t0(T_CONNECT); if (feat_trace == FEAT_FULL) connect_t<FEAT_FULL>(it); else if (feat_trace == FEAT_IMG_ENV) connect_t<FEAT_IMG_ENV>(it); else connect_t<FEAT_SIMPLE>(it); t1(); }
    // The Sobol' dimensions a vertex of round `it` can draw: a path starts the round at dimension <= 3 + 8 it (two camera
    // dimensions, the skipped dimension 4, at most 8 per vertex before) and draws at most 9 further ones; the window staged
    // into LDS is the top of that range (paths below it -- long specular chains -- read the global tables).
    ShadeLdsCfg shade_cfg(uint32_t it) const {
        ShadeLdsCfg c;
        c.sob_nib = (2u * S.log2_res + (31u - (uint32_t)__builtin_clz(S.spp)) <= 32u) ? 8u : (uint32_t)SOBOL_NIBBLES;
        const uint32_t cap = SH_SOB_WORDS / sob_stride(c.sob_nib), top = std::min<uint32_t>(1024u, 3u + 8u * it + 9u);
        c.sob_lo = top > cap ? top - cap : 0u; c.sob_n = top - c.sob_lo;
        c.tri_lds = (sc.n_prims * SH_TRI_REC_V4 <= SH_TRI_V4 && opt.shade_lds) ? 1u : 0u;
        c.n_lights_lds = opt.shade_lds ? std::min<uint32_t>(sc.n_lights, SH_LIGHTS) : 0u;
        if (!opt.shade_lds) c.sob_n = 0;
        c.pre_li = presample_li();
        c.marg_li = 0xffffffffu;
        if (opt.shade_lds && !ps->H.inf_lights.empty()) {
            const DLight &Le = ps->H.lights[ps->H.inf_lights[0]];
            if (Le.nv >= 1 && (uint32_t)Le.nv <= SH_MARG_N && Le.guide_v >= 1 && Le.guide_v <= SH_MARG_N) c.marg_li = ps->H.inf_lights[0];
        }
        return c;
    }
    typedef void (*ShadeFn)(DParams, DSampler, DCamera, DScene, DPaths, DQueues, uint32_t, uint32_t, ShadeLdsCfg, uint32_t, uint32_t *);
    template <int FEAT, bool ENVPRE> ShadeFn shade_fn_t(int kind) {
        switch (kind) {
            case 0: return k_shade<0, FEAT, ENVPRE>;
            case 1: return k_shade<1, FEAT, ENVPRE>;
            case 2: return k_shade<2, FEAT, ENVPRE>;
            case 3: return k_shade<3, FEAT, ENVPRE>;
            case 4: return k_shade<4, FEAT, ENVPRE>;
            default: return k_shade<5, FEAT, ENVPRE>;
        }
    }
    // (ENVPRE: every InfiniteAreaLight sample of the render comes from k_env_presample -- the scene has one such light and it is the
    // presampled one -- so the shade kernels are built without the walk of its distribution)
    bool env_presampled_only() const { return presample_li() != 0xffffffffu && ps->H.inf_lights.size() == 1; }
    ShadeFn shade_fn(int kind) {
        if (feat == FEAT_SIMPLE) return shade_fn_t<FEAT_SIMPLE, false>(kind);
        if (feat == FEAT_IMG) return shade_fn_t<FEAT_IMG, false>(kind);
        if (feat == FEAT_IMG_ENV) return env_presampled_only() ? shade_fn_t<FEAT_IMG_ENV, true>(kind) : shade_fn_t<FEAT_IMG_ENV, false>(kind);
        return env_presampled_only() ? shade_fn_t<FEAT_FULL, true>(kind) : shade_fn_t<FEAT_FULL, false>(kind);
    }
    // the environment light k_env_presample serves: one InfiniteAreaLight whose marginal tables fit the LDS area, Sobol' sampler
    uint32_t presample_li() const {
        if (!opt.env_presample || !(feat & FEAT_INFINITE) || S.kind != PTRS_SAMPLER_SOBOL || ps->H.inf_lights.empty()) return 0xffffffffu;
        const DLight &Le = ps->H.lights[ps->H.inf_lights[0]];
        return (Le.nv >= 1 && (uint32_t)Le.nv <= SH_MARG_N && Le.guide_v >= 1 && Le.guide_v <= SH_MARG_N) ? ps->H.inf_lights[0] : 0xffffffffu;
    }
    void presample(uint32_t it) {
        const uint32_t li = presample_li();
        const uint32_t nee_kinds = kinds_mask & 0x3fu & ~((1u << PTRS_MAT_MIRROR) | (1u << PTRS_MAT_GLASS));
        if (li == 0xffffffffu || !nee_kinds) return;
        t0(T_AUX);
        if (feat == FEAT_IMG_ENV) hipLaunchKernelGGL((k_env_presample<FEAT_IMG_ENV>), dim3(persistent_grid(k_env_presample<FEAT_IMG_ENV>, T_AUX)), dim3(BLOCK), 0, stream, S, sc, P, Q, it, seg_cap, nee_kinds, li, shade_cfg(it), G, ticket(it, TK_PRESAMPLE));
        else hipLaunchKernelGGL((k_env_presample<FEAT_FULL>), dim3(persistent_grid(k_env_presample<FEAT_FULL>, T_AUX)), dim3(BLOCK), 0, stream, S, sc, P, Q, it, seg_cap, nee_kinds, li, shade_cfg(it), G, ticket(it, TK_PRESAMPLE));
        t1();
    }
    // ---- the fused tail (k_tail) ----
    typedef void (*TailFn)(DParams, DSampler, DCamera, DScene, StackSpill, DPaths, DQueues, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, ShadeLdsCfg, uint32_t, uint32_t *);
    enum : uint32_t { TAIL_PATHS_DEFAULT = 192 }; // measured (tools/tail_sweep.sh): flat between 64 and 384 paths per segment for every job size; below, thin rounds stay three launches each, above, thick rounds run at the tail's two waves per SIMD
    TailFn tail_fn() { // the instantiation for this scene, or null: one material bucket, the lean traversal kernels, the 8-entry stack column, the fused epilogue / resolve
        if (!opt.tail || !opt.fused_epilogue || !opt.fused_resolve || !opt.shade_lds || !opt.persist || (ps->stack_lds != 8 && ps->stack_lds != 9) || feat_trace != FEAT_SIMPLE || S.kind != PTRS_SAMPLER_SOBOL) return nullptr;
        int kind = -1;
        for (int k = 0; k < 7; ++k) if (ps->H.kinds_present[k]) { if (kind >= 0) return nullptr; kind = k; }
        const bool ovf = ps->spill.p != nullptr;
        if (ps->stack_lds == 9) return (kind == PTRS_MAT_MATTE && feat == FEAT_SIMPLE) ? (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 544, false, 9> : nullptr;
        if (kind == PTRS_MAT_MATTE && feat == FEAT_SIMPLE) {
            if (geom4 <= 640) return ovf ? (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 640, true> : (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 640, false>;
            if (geom4 <= 1536) return ovf ? (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 1536, true> : (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 1536, false>;
            return ovf ? (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 0, true> : (TailFn)k_tail<PTRS_MAT_MATTE, FEAT_SIMPLE, 0, false>;
        }
        if (kind == PTRS_MAT_DISNEY && feat == FEAT_IMG && geom4 > 1536) return ovf ? (TailFn)k_tail<PTRS_MAT_DISNEY, FEAT_IMG, 0, true> : (TailFn)k_tail<PTRS_MAT_DISNEY, FEAT_IMG, 0, false>;
        return nullptr;
    }
    // The round at which the current pass (pass_begin has run: G is known) hands over to the tail, or 0xffffffff: the first round whose
    // expected alive paths -- this scene's survival profile x the pass's paths -- are at most tail_paths per segment.  No profile yet
    // (the scene's first render): no tail; the results do not depend on the choice.
    uint32_t tail_round(uint32_t n_paths, uint32_t n_rounds) {
        if (!tail_fn()) return 0xffffffffu;
        if (opt.tail_at >= 0) return (uint32_t)opt.tail_at < n_rounds ? (uint32_t)opt.tail_at : 0xffffffffu;
        const double limit = (double)(opt.tail_paths > 0 ? (uint32_t)opt.tail_paths : (uint32_t)TAIL_PATHS_DEFAULT) * (double)G;
        const std::vector<double> &f = ps->alive_frac;
        for (uint32_t it = 0; it < n_rounds && it < f.size(); ++it) if (f[it] * (double)n_paths <= limit) return it;
        return 0xffffffffu;
    }
    uint32_t tail_at_last = 0xffffffffu;
    void tail(uint32_t it0, uint32_t it_end) {
        const TailFn fn = tail_fn();
        if (!fn) return;
        t0(T_TAIL);
        hipLaunchKernelGGL(fn, dim3(persistent_grid(fn, T_TAIL)), dim3(BLOCK), 0, stream, R, S, C, sc, lane_spill(), P, Q, it0, it_end, seg_cap, refill, refill_connect, kinds_mask & 0x3fu, shade_cfg(it0), G, ticket(it0, TK_TAIL));
        t1();
        tail_at_last = it0;
    }
    void learn(const uint32_t *counts, uint32_t n_rows, uint32_t n_paths) { // counts[row * Q_STRIDE + q] of a finished pass
        if (!n_paths || n_paths < 4096u) return; // (too few paths to speak for the scene)
        std::vector<double> f(n_rows);
        for (uint32_t i = 0; i < n_rows; ++i) f[i] = (double)counts[(size_t)i * Q_STRIDE + Q_EXT] / (double)n_paths;
        ps->alive_frac.swap(f);
    }
    void shade(uint32_t it, int kind) {
        t0(T_SHADE);
        if (kind > 5) kind = 5;
        const ShadeFn fn = shade_fn(kind);
        hipLaunchKernelGGL(fn, dim3(persistent_grid(fn, T_SHADE)), dim3(BLOCK), 0, stream, R, S, C, sc, P, Q, it, seg_cap, shade_cfg(it), G, ticket(it, TK_SHADE0 + kind));
        t1();
    }
    void reduce_counts(uint32_t n_rows) { hipLaunchKernelGGL(k_reduce_counts, dim3(n_rows * Q_STRIDE), dim3(BLOCK), 0, stream, (const uint32_t *)Q.counts, G, (uint32_t *)ps->totals[cur].p); }
    uint32_t read_count(uint32_t it, int q) {
        std::vector<uint32_t> seg(G);
        (void)hipMemcpyAsync(seg.data(), Q.counts + ((size_t)it * Q_STRIDE + (size_t)q) * G, (size_t)G * 4, hipMemcpyDeviceToHost, stream);
        if (hipStreamSynchronize(stream) != hipSuccess) { rc = PTRS_ERR_DEVICE; return 0; }
        uint64_t t = 0; for (uint32_t v : seg) t += v;
        return t > 0xffffffffull ? 0xffffffffu : (uint32_t)t;
    }
    void read_counts(uint32_t *dst, uint32_t n_rows) {
        reduce_counts(n_rows);
        (void)hipMemcpyAsync(dst, ps->totals[cur].p, (size_t)n_rows * Q_STRIDE * 4, hipMemcpyDeviceToHost, stream);
        if (hipStreamSynchronize(stream) != hipSuccess) rc = PTRS_ERR_DEVICE;
    }
    void film(v4 *film_px, int32_t y0, int32_t y1) {
        const int32_t tiles_x = (R.W + 15) / 16, tiles_y = (y1 - y0 + 15) / 16;
        // film kernels run in pass order whichever lane they are on: the film is one running sum per pixel
        if (n_lanes > 1 && film_prev) (void)hipStreamWaitEvent(stream, film_prev, 0);
        t0(T_FILM); hipLaunchKernelGGL(k_film, dim3((uint32_t)tiles_x * (uint32_t)tiles_y), dim3(BLOCK), 0, stream, R, S, P, (const float *)ps->table.p, film_px, y0, y1, tiles_x); t1();
        if (n_lanes > 1) { film_prev = ps->lane_ev[cur]; (void)hipEventRecord(film_prev, stream); }
    }
    // progressive render: wait for this lane's film kernel (and, through the chain of events, every earlier pass's) and copy the rows out
    PtrsFilmPixel *host_film = nullptr; int32_t film_w = 0;
    void publish_rows(v4 *film_px, int32_t y0, int32_t y1) {
        if (!host_film || hipStreamSynchronize(stream) != hipSuccess) { rc = host_film ? PTRS_ERR_DEVICE : rc; return; }
        const size_t off = (size_t)y0 * (size_t)film_w, cnt = (size_t)(y1 - y0) * (size_t)film_w;
        // (on the lane's own stream: a blocking copy on the legacy null stream would also wait for whatever lane 0 has queued behind this pass)
        if (hipMemcpyAsync(host_film + off, film_px + off, cnt * sizeof(PtrsFilmPixel), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) rc = PTRS_ERR_DEVICE;
    }
    void dump_rays(uint32_t it, const RayDump &d) {
        DevBuf cnt;
        if (cnt.ensure(16) != PTRS_OK) { rc = PTRS_ERR_DEVICE; return; }
        (void)hipMemsetAsync(cnt.p, 0, 16, stream);
        hipLaunchKernelGGL(k_dump_rays, dim3((G + WAVES - 1) / WAVES), dim3(BLOCK), 0, stream, P, Q, it, seg_cap, G, d.max_rays, d.out, (uint32_t *)cnt.p);
        uint32_t n = 0;
        if (hipMemcpyAsync(&n, cnt.p, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) rc = PTRS_ERR_DEVICE;
        *d.n_out = n < d.max_rays ? n : d.max_rays;
        cnt.release();
    }
    void export_samples(float *out) { t0(T_FILM); hipLaunchKernelGGL(k_export_samples, dim3(grid_for(R.n_paths)), dim3(BLOCK), 0, stream, R, S, P, out); t1(); }
    void end(PtrsStats &st) {
        for (uint32_t l = 0; l < n_lanes; ++l) { select(l); if (hipStreamSynchronize(stream) != hipSuccess) rc = PTRS_ERR_DEVICE; }
        select(0);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { g_err = std::string("kernel launch failed: ") + hipGetErrorString(le); rc = PTRS_ERR_DEVICE; }
        unsigned long long hs[CNT_NUM] = {0};
        (void)hipMemcpy(hs, Q.stats, sizeof(hs), hipMemcpyDeviceToHost);
        st.nodes_visited = hs[CNT_NODES]; st.tris_tested = hs[CNT_TRIS]; st.node_steps_x64 = hs[CNT_NODE_STEPS]; st.node_visits = hs[CNT_NODE_VISITS]; st.tri_steps_x64 = hs[CNT_TRI_STEPS]; st.error_flags = hs[CNT_ERR];
        for (int k = 0; k < 12; ++k) st.debug[k] = hs[CNT_STAMP0 + k];
        st.kernel_launches = launches; st.trace_launches = trace_launches;
        for (auto &s : spans) {
            float ms = 0.0f;
            if (s.a && s.b && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
                if (s.cat == T_EXTEND) st.ms_extend += ms; else if (s.cat == T_CONNECT) st.ms_connect += ms; else if (s.cat == T_SHADE) st.ms_shade_kernels += ms; else if (s.cat == T_AUX) st.ms_aux += ms; else if (s.cat == T_TAIL) st.ms_tail += ms; else st.ms_film += ms;
            }
        }
        st.ms_trace = st.ms_extend + st.ms_connect; st.ms_shade = st.ms_shade_kernels + st.ms_aux;
        st.queue_segments = G; st.lanes = n_lanes; st.grid_pct = (uint64_t)(opt.grid_pct ? opt.grid_pct : 100);
        { const int cls[4] = {T_EXTEND, T_CONNECT, T_SHADE, T_AUX}; for (int k = 0; k < 4; ++k) { st.grid_wgs[k] = last_grid[cls[k]]; st.resident_wgs_per_cu[k] = last_per_cu[cls[k]]; } }
        st.extend_launches = cat_launches[T_EXTEND]; st.connect_launches = cat_launches[T_CONNECT]; st.shade_launches = cat_launches[T_SHADE]; st.aux_launches = cat_launches[T_AUX]; st.film_launches = cat_launches[T_FILM]; st.tail_launches = cat_launches[T_TAIL]; st.tail_round = tail_at_last;
        size_t bytes = 0;
        for (auto &l : ps->ws) for (auto &b : l) bytes += b.bytes;
        for (auto &b : ps->counts) bytes += b.bytes;
        st.device_bytes = bytes + ps->film_tmp.bytes + ps->samples_tmp.bytes + ps->stack_spill.bytes + ps->row_cost.bytes;
    }
};

Options scene_options(const PtrsScene *ps) { // the process-wide knobs with this scene's overrides on top
    Options o = options();
    for (const OptionDesc &d : k_options) { auto it = ps->overrides.find(d.name); if (it != ps->overrides.end()) o.*(d.field) = it->second; }
    return o;
}

int do_render(PtrsScene *ps, const PtrsCamera *cam, const PtrsRenderParams *prm, v4 *film_dev, float *samples_dev, hipStream_t stream, PtrsStats *stats, const int32_t *single_pixel = nullptr,
              const RenderProgress *progress = nullptr, PtrsFilmPixel *host_film = nullptr, int share = 1, const RayDump *dump = nullptr, uint32_t *row_cost_dev = nullptr) {
    if (!ps || !cam || !prm || (!film_dev && !single_pixel && !row_cost_dev)) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    HIPCHK(hipSetDevice(ps->device));
    HipBackend be;
    be.ps = ps; be.stream = stream; be.opt = scene_options(ps); be.host_film = host_film; be.film_w = prm->width; be.share = share;
    if (dump && be.opt.lanes == 0) be.opt.lanes = 1; // a ray dump is of the render's first pass: one lane gives that pass as many of the job's paths as memory holds
    int rc = get_sobol(ps->device, &be.sob);
    if (rc != PTRS_OK) return rc;
    std::string err;
    rc = render_impl(be, ps->sc, ps->H, ps->H.max_depth, *cam, *prm, film_dev, samples_dev, stats, err, progress, dump, single_pixel, row_cost_dev);
    if (rc != PTRS_OK) { if (!err.empty()) g_err = err; return rc; }
    if (be.rc != PTRS_OK) { if (g_err.empty()) g_err = "device error during render"; return be.rc; }
    return PTRS_OK;
}

// extern "C" entry points never let an exception (std::bad_alloc from a host vector, ...) cross the boundary
template <class F> int guarded(F &&f) {
    try { return f(); }
    catch (const std::bad_alloc &) { g_err = "out of host memory"; return PTRS_ERR_DEVICE; }
    catch (const std::exception &e) { g_err = std::string("internal error: ") + e.what(); return PTRS_ERR_INVALID; }
    catch (...) { g_err = "internal error"; return PTRS_ERR_INVALID; }
}

} // namespace

extern "C" {

int ptrs_abi_version(void) { return PTRS_ABI_VERSION; }
#ifndef PTRS_BUILD_ID
#define PTRS_BUILD_ID "unknown"
#endif
const char *ptrs_build_id(void) { return PTRS_BUILD_ID; }
const char *ptrs_last_error(void) { return g_err.c_str(); }

// struct sizes as compiled, for the binding self-check (tests/test_abi.py)
int ptrs_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(PtrsTexture); case 1: return (int)sizeof(PtrsMaterial); case 2: return (int)sizeof(PtrsMesh);
        case 3: return (int)sizeof(PtrsLight); case 4: return (int)sizeof(PtrsBvhNode); case 5: return (int)sizeof(PtrsSceneDesc);
        case 6: return (int)sizeof(PtrsCamera); case 7: return (int)sizeof(PtrsRenderParams); case 8: return (int)sizeof(PtrsStats);
        case 9: return (int)sizeof(PtrsHit); case 10: return (int)sizeof(PtrsFilmPixel);
        default: return -1;
    }
}

int ptrs_set_option(const char *name, int64_t value) {
    if (!name) { g_err = "null option name"; return PTRS_ERR_INVALID; }
    for (const OptionDesc &o : k_options)
        if (!std::strcmp(name, o.name)) {
            if (value < o.lo || value > o.hi) { g_err = std::string("option ") + name + ": value outside [" + std::to_string(o.lo) + ", " + std::to_string(o.hi) + "]"; return PTRS_ERR_INVALID; }
            std::lock_guard<std::mutex> lk(g_opt_mu);
            g_opt.*(o.field) = (int)value;
            return PTRS_OK;
        }
    g_err = std::string("unknown option ") + name;
    return PTRS_ERR_INVALID;
}

int ptrs_get_option(const char *name, int64_t *value) {
    if (!name || !value) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    for (const OptionDesc &o : k_options)
        if (!std::strcmp(name, o.name)) { std::lock_guard<std::mutex> lk(g_opt_mu); *value = g_opt.*(o.field); return PTRS_OK; }
    g_err = std::string("unknown option ") + name;
    return PTRS_ERR_INVALID;
}

static int scene_create_impl(const PtrsSceneDesc *desc, int32_t device, PtrsScene **out);
int ptrs_scene_create(const PtrsSceneDesc *desc, int32_t device, PtrsScene **out) { return guarded([&]() { return scene_create_impl(desc, device, out); }); }
static int scene_create_impl(const PtrsSceneDesc *desc, int32_t device, PtrsScene **out) {
    if (!desc || !out) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (this library has no CPU fallback)"; return PTRS_ERR_DEVICE; }
    if (device < 0 || device >= ndev) { g_err = "device ordinal out of range"; return PTRS_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    PtrsScene *ps = new PtrsScene();
    ps->device = device;
    const Options opt = options();
    int rc = build_host_scene(*desc, ps->H, g_err, opt.node_form, opt.node_order);
    if (rc != PTRS_OK) { delete ps; return rc; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ps->n_cu = prop.multiProcessorCount;
    HostScene &H = ps->H;
    if ((rc = upload(ps->nodes2, H.nodes2)) || (rc = upload(ps->nodes4, H.nodes4)) || (rc = upload(ps->nodes, H.nodes)) || (rc = upload(ps->tris, H.tris)) || (rc = upload(ps->shade, H.shade)) || (rc = upload(ps->mats, H.mats)) ||
        (rc = upload(ps->texs, H.texs)) || (rc = upload(ps->levels, H.levels)) || (rc = upload(ps->texdata, H.texdata)) || (rc = upload(ps->lights, H.lights)) ||
        (rc = upload(ps->distdata, H.distdata)) || (rc = upload(ps->inf, H.inf_lights))) { delete ps; return rc; }
    DScene &sc = ps->sc;
    sc.nodes2 = (const DNode2 *)ps->nodes2.p; sc.n_nodes2 = (uint32_t)H.nodes2.size(); sc.nodes4 = (const DNode4 *)ps->nodes4.p; sc.n_nodes4 = (uint32_t)H.nodes4.size();
    sc.nodes = (const DNode *)ps->nodes.p; sc.tris = (const DTri *)ps->tris.p; sc.shade = (const DTriShade *)ps->shade.p; sc.mats = (const DMaterial *)ps->mats.p;
    sc.texs = (const DTexture *)ps->texs.p; sc.levels = (const DTexLevel *)ps->levels.p; sc.texdata = (const float *)ps->texdata.p; sc.lights = (const DLight *)ps->lights.p;
    sc.distdata = (const float *)ps->distdata.p; sc.inf_lights = (const uint32_t *)ps->inf.p;
    sc.n_nodes = (uint32_t)H.nodes.size(); sc.n_prims = (uint32_t)H.tris.size(); sc.n_lights = (uint32_t)H.lights.size(); sc.n_inf = (uint32_t)H.inf_lights.size();
    // traversal stack: 8 LDS entries per lane; what a deeper tree can stack beyond that spills to a global column per
    // resident thread.  Quad-form scenes spend the LDS this saves on the top QUAD_TOP_NODES records of the tree (measured
    // against a 16-entry column without the cache: +2 % on colonnade, +1 % on classroom).  the option stack_lds = 16 selects that
    // older layout for quad-form scenes.
    ps->stack_lds = (H.use_quad && opt.stack_lds == 16) ? 16u : 8u;
    if (!H.use_quad && opt.stack_lds != 16 && H.stack_bound == 9u && LN_V4 * (uint32_t)H.nodes2.size() + 9u * (uint32_t)H.tris.size() <= 544u) ps->stack_lds = 9u; // the 9 / 544 LDS form (TravWaves): no overflow code
    if (H.stack_bound > ps->stack_lds) {
        const size_t threads = (size_t)ps->n_cu * 8 * BLOCK; // no launch holds more than 8 workgroups per CU
        ps->spill_lane_elems = threads * (size_t)(H.stack_bound - ps->stack_lds);
        if ((rc = ps->stack_spill.ensure(ps->spill_lane_elems * sizeof(unsigned long long))) != PTRS_OK) { delete ps; return rc; } // one lane's worth (ptrs_trace_rays / ptrs_trace_bench); a render grows it to its lanes (HipBackend::begin)
        ps->spill.p = (unsigned long long *)ps->stack_spill.p; ps->spill.stride = (uint32_t)threads;
    }
    *out = ps;
    return PTRS_OK;
}

void ptrs_scene_destroy(PtrsScene *scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    delete scene;
}

int ptrs_scene_info(PtrsScene *scene, uint64_t *n_nodes, uint64_t *max_depth, uint64_t *n_tris) {
    if (!scene) { g_err = "null scene"; return PTRS_ERR_INVALID; }
    if (n_nodes) *n_nodes = scene->H.nodes.size();
    if (max_depth) *max_depth = scene->H.max_depth;
    if (n_tris) *n_tris = scene->H.tris.size();
    return PTRS_OK;
}

int ptrs_render_device(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, void *film_inout_device, void *hip_stream, PtrsStats *stats) {
    return guarded([&]() { return do_render(scene, camera, params, (v4 *)film_inout_device, nullptr, (hipStream_t)hip_stream, stats); });
}

static int render_samples_impl(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, PtrsFilmPixel *film_inout, float *sample_rgb, PtrsStats *stats);
int ptrs_render_samples(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, PtrsFilmPixel *film_inout, float *sample_rgb, PtrsStats *stats) {
    return guarded([&]() { return render_samples_impl(scene, camera, params, film_inout, sample_rgb, stats); });
}
static int render_samples_impl(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, PtrsFilmPixel *film_inout, float *sample_rgb, PtrsStats *stats) {
    if (!scene || !params || !film_inout) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    HIPCHK(hipSetDevice(scene->device));
    const size_t npx = (size_t)params->width * (size_t)params->height;
    int rc = scene->film_tmp.ensure(npx * sizeof(PtrsFilmPixel));
    if (rc != PTRS_OK) return rc;
    // only the rows of the band travel (rows outside [row_begin, row_end) are untouched: concurrent renders of disjoint
    // bands into one host film, one per device, do not step on each other)
    int32_t rb = params->row_begin, re = params->row_end;
    if (re <= rb) { rb = 0; re = params->height; }
    if (rb < 0 || re > params->height) { g_err = "row band outside the film"; return PTRS_ERR_INVALID; }
    const size_t band_off = (size_t)rb * (size_t)params->width, band_px = (size_t)(re - rb) * (size_t)params->width;
    HIPCHK(hipMemcpy((PtrsFilmPixel *)scene->film_tmp.p + band_off, film_inout + band_off, band_px * sizeof(PtrsFilmPixel), hipMemcpyHostToDevice));
    float *sdev = nullptr; size_t sbytes = 0;
    if (sample_rgb) {
        const SampleGrid g = make_sample_grid(params->width, params->height, params->spp);
        const size_t spp = params->sampler == PTRS_SAMPLER_STRATIFIED ? (size_t)std::max(params->spp, 1) : (size_t)g.spp; // the stratified sampler takes spp = dim^2 as is
        sbytes = (size_t)g.NX * (size_t)g.NY * spp * 3 * sizeof(float);
        if ((rc = scene->samples_tmp.ensure(sbytes)) != PTRS_OK) return rc;
        HIPCHK(hipMemset(scene->samples_tmp.p, 0, sbytes));
        sdev = (float *)scene->samples_tmp.p;
    }
    rc = do_render(scene, camera, params, (v4 *)scene->film_tmp.p, sdev, nullptr, stats);
    if (rc != PTRS_OK) return rc;
    HIPCHK(hipMemcpy(film_inout + band_off, (PtrsFilmPixel *)scene->film_tmp.p + band_off, band_px * sizeof(PtrsFilmPixel), hipMemcpyDeviceToHost));
    if (sample_rgb) HIPCHK(hipMemcpy(sample_rgb, sdev, sbytes, hipMemcpyDeviceToHost));
    return PTRS_OK;
}

int ptrs_render(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, PtrsFilmPixel *film_inout, PtrsStats *stats) {
    return ptrs_render_samples(scene, camera, params, film_inout, nullptr, stats);
}

int ptrs_render_progressive(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, PtrsFilmPixel *film_inout, PtrsProgressFn fn, void *user, PtrsStats *stats) {
    return guarded([&]() -> int {
        if (!scene || !params || !film_inout) { g_err = "null argument"; return PTRS_ERR_INVALID; }
        HIPCHK(hipSetDevice(scene->device));
        int32_t rb = params->row_begin, re = params->row_end;
        if (re <= rb) { rb = 0; re = params->height; }
        if (rb < 0 || re > params->height) { g_err = "row band outside the film"; return PTRS_ERR_INVALID; }
        const size_t npx = (size_t)params->width * (size_t)params->height;
        int rc = scene->film_tmp.ensure(npx * sizeof(PtrsFilmPixel));
        if (rc != PTRS_OK) return rc;
        const size_t off = (size_t)rb * (size_t)params->width, cnt = (size_t)(re - rb) * (size_t)params->width;
        HIPCHK(hipMemcpy((PtrsFilmPixel *)scene->film_tmp.p + off, film_inout + off, cnt * sizeof(PtrsFilmPixel), hipMemcpyHostToDevice));
        const RenderProgress pg{fn, user};
        rc = do_render(scene, camera, params, (v4 *)scene->film_tmp.p, nullptr, nullptr, stats, nullptr, &pg, film_inout);
        if (rc != PTRS_OK) return rc;
        HIPCHK(hipMemcpy(film_inout + off, (PtrsFilmPixel *)scene->film_tmp.p + off, cnt * sizeof(PtrsFilmPixel), hipMemcpyDeviceToHost));
        return PTRS_OK;
    });
}

// ---- one process, N devices ---------------------------------------------------------------------------
int ptrs_plan_bands(int32_t height, uint32_t n, const float *row_cost, int32_t *bounds_out) {
    if (height <= 0 || n == 0 || !bounds_out) { g_err = "bad band request"; return PTRS_ERR_INVALID; }
    return guarded([&]() -> int { plan_bands(height, n, row_cost, bounds_out); return PTRS_OK; });
}

// Band planning probe: `params`' render (the caller gives spp = 1) with one counter per sample row on the device -- every BVH
// query (extension, shadow, MIS ray) of a path is added to its row when the stage that issues it runs -- and no film.  One call of a few
// milliseconds where the planner of round 3 made 64 strip renders (270 ms for Cornell).  row_cost_out[y], y in [0, height): the queries of
// the sample row under output row y (grid row y + 2); the apron's rows are left out.
int ptrs_render_row_cost(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, float *row_cost_out, PtrsStats *stats) {
    return guarded([&]() -> int {
        if (!scene || !camera || !params || !row_cost_out) { g_err = "null argument"; return PTRS_ERR_INVALID; }
        if (params->width <= 0 || params->height <= 0) { g_err = "bad render parameters"; return PTRS_ERR_INVALID; }
        HIPCHK(hipSetDevice(scene->device));
        const SampleGrid g = make_sample_grid(params->width, params->height, params->spp);
        int rc = scene->row_cost.ensure((size_t)g.NY * 4);
        if (rc != PTRS_OK) return rc;
        HIPCHK(hipMemset(scene->row_cost.p, 0, (size_t)g.NY * 4));
        PtrsRenderParams p = *params;
        p.row_begin = 0; p.row_end = p.height;
        rc = do_render(scene, camera, &p, nullptr, nullptr, nullptr, stats, nullptr, nullptr, nullptr, 1, nullptr, (uint32_t *)scene->row_cost.p);
        if (rc != PTRS_OK) return rc;
        std::vector<uint32_t> h((size_t)g.NY);
        HIPCHK(hipMemcpy(h.data(), scene->row_cost.p, (size_t)g.NY * 4, hipMemcpyDeviceToHost));
        for (int32_t y = 0; y < params->height; ++y) row_cost_out[y] = (float)h[(size_t)(y - g.min_y)];
        return PTRS_OK;
    });
}

int ptrs_scene_set_option(PtrsScene *scene, const char *name, int64_t value) {
    if (!scene || !name) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    for (const OptionDesc &o : k_options)
        if (!std::strcmp(name, o.name)) {
            if (value < o.lo || value > o.hi) { g_err = std::string("option ") + name + ": value outside [" + std::to_string(o.lo) + ", " + std::to_string(o.hi) + "]"; return PTRS_ERR_INVALID; }
            return guarded([&]() -> int { scene->overrides[name] = (int)value; return PTRS_OK; });
        }
    g_err = std::string("unknown option ") + name;
    return PTRS_ERR_INVALID;
}

int ptrs_render_multi(PtrsScene *const *scenes, uint32_t n, const PtrsCamera *camera, const PtrsRenderParams *params, const int32_t *band_bounds,
                      PtrsFilmPixel *film_inout, PtrsStats *stats_per_scene) {
    return guarded([&]() -> int {
        if (!scenes || n == 0 || !camera || !params || !film_inout) { g_err = "null argument"; return PTRS_ERR_INVALID; }
        for (uint32_t i = 0; i < n; ++i) {
            if (!scenes[i]) { g_err = "null scene"; return PTRS_ERR_INVALID; }
            for (uint32_t j = 0; j < i; ++j) if (scenes[j] == scenes[i]) { g_err = "the same scene handle twice: every band needs a scene (workspace, film) of its own"; return PTRS_ERR_INVALID; }
        }
        const int32_t H = params->height, W = params->width;
        std::vector<int32_t> bounds(n + 1);
        if (band_bounds) { bounds.assign(band_bounds, band_bounds + n + 1); }
        else plan_bands(H, n, nullptr, bounds.data());
        if (bounds[0] != 0 || bounds[n] != H) { g_err = "band bounds must run from 0 to height"; return PTRS_ERR_INVALID; }
        for (uint32_t i = 0; i < n; ++i) if (bounds[i + 1] < bounds[i]) { g_err = "band bounds must not decrease"; return PTRS_ERR_INVALID; }
        // Each device renders its band into a band-sized film of its own and, as soon as its render is through, pushes the band into
        // device 0's film over xGMI (hipMemcpyPeerAsync on a stream of its own; staged through the host film where two devices cannot
        // reach each other): no device waits for another, the gather overlaps the slower devices' renders.  Device 0's film leaves
        // through one device-to-host copy.  Rows are disjoint and every sample's value depends only on (pixel, sample index), so the
        // result is bit-identical to ptrs_render on one device.
        const size_t npx = (size_t)W * (size_t)H;
        PtrsScene *root = scenes[0];
        HIPCHK(hipSetDevice(root->device));
        int rc = root->film_tmp.ensure(npx * sizeof(PtrsFilmPixel));
        if (rc != PTRS_OK) return rc;
        const bool film_is_zero = (params->flags & PTRS_FLAG_FILM_ZERO) != 0; // the caller vouches for a cleared film: nothing to upload
        std::vector<int> share(n, 0);
        for (uint32_t i = 0; i < n; ++i) for (uint32_t j = 0; j < n; ++j) if (scenes[j]->device == scenes[i]->device && bounds[j + 1] > bounds[j]) ++share[i]; // renders dividing one device's memory
        std::vector<int> rcs(n, PTRS_OK); std::vector<std::string> errs(n);
        std::vector<char> staged(n, 0); // band i travelled through film_inout (no peer path): it is copied into the root film after the join
        auto work = [&](uint32_t i) {
            PtrsScene *ps = scenes[i];
            rcs[i] = guarded([&]() -> int {
                if (stats_per_scene) std::memset(&stats_per_scene[i], 0, sizeof(PtrsStats));
                if (bounds[i + 1] == bounds[i]) return PTRS_OK;
                HIPCHK(hipSetDevice(ps->device));
                const size_t off = (size_t)bounds[i] * W, cnt = (size_t)(bounds[i + 1] - bounds[i]) * W, bytes = cnt * sizeof(PtrsFilmPixel);
                // the band's rows only; the render addresses the film by absolute row, so it gets the band's base moved back by `off`
                PtrsFilmPixel *band = nullptr;
                if (ps == root) band = (PtrsFilmPixel *)root->film_tmp.p + off;
                else { int rc2 = ps->film_tmp.ensure(bytes); if (rc2 != PTRS_OK) return rc2; band = (PtrsFilmPixel *)ps->film_tmp.p; }
                if (film_is_zero) HIPCHK(hipMemset(band, 0, bytes)); else HIPCHK(hipMemcpy(band, film_inout + off, bytes, hipMemcpyHostToDevice));
                PtrsRenderParams p = *params;
                p.row_begin = bounds[i]; p.row_end = bounds[i + 1]; p.device = ps->device;
                int rc2 = do_render(ps, camera, &p, (v4 *)(band - off), nullptr, nullptr, stats_per_scene ? &stats_per_scene[i] : nullptr, nullptr, nullptr, nullptr, std::max(1, share[i]));
                if (rc2 != PTRS_OK || ps == root) return rc2;
                PtrsFilmPixel *dst = (PtrsFilmPixel *)root->film_tmp.p + off;
                const bool direct = scene_options(ps).peer_copy != 0; // (0: the test hook that forces the host-staged path, also between replicas on one device)
                if (direct && ps->device == root->device) { HIPCHK(hipMemcpy(dst, band, bytes, hipMemcpyDeviceToDevice)); return PTRS_OK; }
                bool pushed = false;
                if (direct) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, ps->device, root->device) == hipSuccess && can) {
                        const hipError_t pe = hipDeviceEnablePeerAccess(root->device, 0);
                        if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                        else (void)hipGetLastError();
                    }
                    if (!ps->lane_stream[0] && hipStreamCreateWithFlags(&ps->lane_stream[0], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); }
                    if (ps->lane_stream[0] && hipMemcpyPeerAsync(dst, root->device, band, ps->device, bytes, ps->lane_stream[0]) == hipSuccess && hipStreamSynchronize(ps->lane_stream[0]) == hipSuccess) pushed = true;
                    else (void)hipGetLastError();
                }
                if (!pushed) { HIPCHK(hipMemcpy(film_inout + off, band, bytes, hipMemcpyDeviceToHost)); staged[i] = 1; } // no peer path: through the host film
                return PTRS_OK;
            });
            if (rcs[i] != PTRS_OK) errs[i] = g_err; // thread-local message of the worker
        };
        std::vector<std::thread> th;
        try {
            for (uint32_t i = 1; i < n; ++i) th.emplace_back(work, i);
        } catch (...) {
            for (auto &t : th) if (t.joinable()) t.join();
            g_err = "cannot start a host thread per device";
            return PTRS_ERR_DEVICE;
        }
        work(0); // device 0's band on the calling thread
        for (auto &t : th) t.join();
        for (uint32_t i = 0; i < n; ++i) if (rcs[i] != PTRS_OK) { g_err = "device " + std::to_string(scenes[i]->device) + ": " + errs[i]; return rcs[i]; }
        HIPCHK(hipSetDevice(root->device));
        for (uint32_t i = 1; i < n; ++i) if (staged[i]) {
            const size_t off = (size_t)bounds[i] * W, bytes = (size_t)(bounds[i + 1] - bounds[i]) * W * sizeof(PtrsFilmPixel);
            HIPCHK(hipMemcpy((PtrsFilmPixel *)root->film_tmp.p + off, film_inout + off, bytes, hipMemcpyHostToDevice));
        }
        HIPCHK(hipMemcpy(film_inout, root->film_tmp.p, npx * sizeof(PtrsFilmPixel), hipMemcpyDeviceToHost));
        return PTRS_OK;
    });
}

int ptrs_render_single_pixel(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, int32_t px, int32_t py, float *rgb_out) {
    // render_single_pixel (integrator.rs:505-534): the sampler is started on `pixel` and li() evaluated for each of its samples;
    // the reference accepts any pixel (it never touches the film), here: any pixel of the sample bounds, apron included.  Only
    // that pixel's spp paths are traced.
    return guarded([&]() -> int {
        if (!scene || !params || !rgb_out) { g_err = "null argument"; return PTRS_ERR_INVALID; }
        HIPCHK(hipSetDevice(scene->device));
        const SampleGrid g = make_sample_grid(params->width, params->height, params->spp);
        const size_t bytes = (size_t)g.spp * 3 * sizeof(float);
        int rc = scene->samples_tmp.ensure(bytes);
        if (rc != PTRS_OK) return rc;
        HIPCHK(hipMemset(scene->samples_tmp.p, 0, bytes));
        const int32_t pixel[2] = {px, py};
        rc = do_render(scene, camera, params, nullptr, (float *)scene->samples_tmp.p, nullptr, nullptr, pixel);
        if (rc != PTRS_OK) return rc;
        HIPCHK(hipMemcpy(rgb_out, scene->samples_tmp.p, bytes, hipMemcpyDeviceToHost));
        return PTRS_OK;
    });
}

static int trace_rays_impl(PtrsScene *scene, uint32_t n, const float *rays, int32_t any_hit, PtrsHit *hits_out, PtrsStats *stats);
int ptrs_trace_rays(PtrsScene *scene, uint32_t n, const float *rays, int32_t any_hit, PtrsHit *hits_out, PtrsStats *stats) {
    return guarded([&]() { return trace_rays_impl(scene, n, rays, any_hit, hits_out, stats); });
}
static int trace_rays_impl(PtrsScene *scene, uint32_t n, const float *rays, int32_t any_hit, PtrsHit *hits_out, PtrsStats *stats) {
    if (!scene || (n && (!rays || !hits_out))) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    if (scene->H.max_depth > 64) { g_err = "BVH deeper than 64"; return PTRS_ERR_UNSUPPORTED; }
    HIPCHK(hipSetDevice(scene->device));
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return PTRS_OK;
    std::vector<v4> ro(n), rd(n);
    for (uint32_t i = 0; i < n; ++i) { ro[i].x = rays[7 * i]; ro[i].y = rays[7 * i + 1]; ro[i].z = rays[7 * i + 2]; ro[i].w = rays[7 * i + 6]; rd[i].x = rays[7 * i + 3]; rd[i].y = rays[7 * i + 4]; rd[i].z = rays[7 * i + 5]; rd[i].w = 0.0f; }
    DevBuf bo, bd, bh, bc, bs, bt;
    int rc;
    if ((rc = upload(bo, ro)) || (rc = upload(bd, rd)) || (rc = bh.ensure((size_t)n * 16)) || (rc = bc.ensure(16)) || (rc = bs.ensure(CNT_NUM * 8)) || (rc = bt.ensure((size_t)n * 4))) { bo.release(); bd.release(); bh.release(); bc.release(); bs.release(); bt.release(); return rc; }
    hipError_t e = hipMemcpy(bc.p, &n, 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(bs.p, 0, CNT_NUM * 8);
    const uint32_t gmax = (uint32_t)scene->n_cu * 8u; // the spill columns are sized for this many workgroups
    dim3 g((n + BLOCK - 1) / BLOCK > gmax ? gmax : (n + BLOCK - 1) / BLOCK), b(BLOCK);
    hipEvent_t ea, eb; (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
    (void)hipEventRecord(ea, nullptr);
#define LAUNCH_T(ANYV)                                                                                                                                  \
    do {                                                                                                                                                \
        const StackSpill sp = scene->spill;                                                                                                             \
        if (scene->stack_lds == 8 && !sp.p) hipLaunchKernelGGL((k_trace<ANYV, 8, false>), g, b, 0, nullptr, scene->sc, sp, (const uint32_t *)nullptr, (const uint32_t *)bc.p, (const v4 *)bo.p, (const v4 *)bd.p, (u4 *)bh.p, (uint32_t *)bh.p, (float *)bt.p, (unsigned long long *)bs.p, 1u); \
        else if (scene->stack_lds == 8) hipLaunchKernelGGL((k_trace<ANYV, 8, true>), g, b, 0, nullptr, scene->sc, sp, (const uint32_t *)nullptr, (const uint32_t *)bc.p, (const v4 *)bo.p, (const v4 *)bd.p, (u4 *)bh.p, (uint32_t *)bh.p, (float *)bt.p, (unsigned long long *)bs.p, 1u); \
        else if (!sp.p) hipLaunchKernelGGL((k_trace<ANYV, 16, false>), g, b, 0, nullptr, scene->sc, sp, (const uint32_t *)nullptr, (const uint32_t *)bc.p, (const v4 *)bo.p, (const v4 *)bd.p, (u4 *)bh.p, (uint32_t *)bh.p, (float *)bt.p, (unsigned long long *)bs.p, 1u); \
        else hipLaunchKernelGGL((k_trace<ANYV, 16, true>), g, b, 0, nullptr, scene->sc, sp, (const uint32_t *)nullptr, (const uint32_t *)bc.p, (const v4 *)bo.p, (const v4 *)bd.p, (u4 *)bh.p, (uint32_t *)bh.p, (float *)bt.p, (unsigned long long *)bs.p, 1u); \
    } while (0)
    if (any_hit) LAUNCH_T(true); else LAUNCH_T(false);
#undef LAUNCH_T
    (void)hipEventRecord(eb, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipGetLastError();
    float ms = 0.0f; (void)hipEventElapsedTime(&ms, ea, eb);
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (e == hipSuccess) {
        if (any_hit) {
            std::vector<uint32_t> oc(n);
            e = hipMemcpy(oc.data(), bh.p, (size_t)n * 4, hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < n; ++i) { hits_out[i].prim = oc[i] ? 0 : -1; hits_out[i].t = rays[7 * i + 6]; hits_out[i].b0 = hits_out[i].b1 = hits_out[i].b2 = 0.0f; }
        } else {
            std::vector<u4> hh(n); std::vector<float> tt(n);
            e = hipMemcpy(hh.data(), bh.p, (size_t)n * 16, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(tt.data(), bt.p, (size_t)n * 4, hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < n; ++i) { hits_out[i].prim = (int32_t)hh[i].x; hits_out[i].b0 = u2f(hh[i].y); hits_out[i].b1 = u2f(hh[i].z); hits_out[i].b2 = u2f(hh[i].w); hits_out[i].t = tt[i]; }
        }
    }
    if (e == hipSuccess && stats) {
        unsigned long long hs[CNT_NUM];
        e = hipMemcpy(hs, bs.p, sizeof(hs), hipMemcpyDeviceToHost);
        stats->nodes_visited = hs[CNT_NODES]; stats->tris_tested = hs[CNT_TRIS]; stats->ms_trace = ms; stats->trace_launches = 1; stats->kernel_launches = 1;
        if (any_hit) stats->rays_shadow = n; else stats->rays_extension = n;
    }
    bo.release(); bd.release(); bh.release(); bc.release(); bs.release(); bt.release();
    if (e != hipSuccess) { g_err = std::string("ptrs_trace_rays: ") + hipGetErrorString(e); return PTRS_ERR_DEVICE; }
    return PTRS_OK;
}

int ptrs_render_dump_rays(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, uint32_t round, uint32_t max_rays, float *rays_out, uint32_t *n_out) {
    return guarded([&]() -> int {
        if (!scene || !camera || !params || !rays_out || !n_out || max_rays == 0) { g_err = "null argument"; return PTRS_ERR_INVALID; }
        HIPCHK(hipSetDevice(scene->device));
        DevBuf out;
        int rc = out.ensure((size_t)max_rays * 7 * sizeof(float));
        if (rc != PTRS_OK) return rc;
        const size_t npx = (size_t)params->width * (size_t)params->height;
        if ((rc = scene->film_tmp.ensure(npx * sizeof(PtrsFilmPixel))) != PTRS_OK) { out.release(); return rc; }
        const RayDump d{round, max_rays, (float *)out.p, n_out};
        *n_out = 0;
        rc = do_render(scene, camera, params, (v4 *)scene->film_tmp.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, &d);
        if (rc == PTRS_OK && *n_out && hipMemcpy(rays_out, out.p, (size_t)*n_out * 7 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) { g_err = "copying the rays out failed"; rc = PTRS_ERR_DEVICE; }
        out.release();
        return rc;
    });
}

// Traversal alone, the way a frame runs it: the rays are uploaded once, laid out as one pass's path state with an extension queue in
// segments (64-ray blocks dealt round-robin, like k_generate's), and the lane-refill extension kernel of this scene (phase voting,
// LDS-cached top / LDS form, persistent waves with tickets; no epilogue) is launched `repeats` times with an event pair each.
int ptrs_trace_bench(PtrsScene *scene, uint32_t n, const float *rays, uint32_t repeats, PtrsHit *hits_out, PtrsStats *stats) {
    return guarded([&]() -> int {
        if (!scene || !rays || n == 0 || repeats == 0 || !stats) { g_err = "null argument"; return PTRS_ERR_INVALID; }
        if (scene->H.max_depth > 64) { g_err = "BVH deeper than 64"; return PTRS_ERR_UNSUPPORTED; }
        HIPCHK(hipSetDevice(scene->device));
        std::memset(stats, 0, sizeof(*stats));
        HipBackend be;
        be.ps = scene; be.stream = nullptr; be.opt = scene_options(scene); be.sc = scene->sc;
        be.feat_trace = scene_trace_features(scene->H);
        be.vote = be.opt.vote >= 0 ? be.opt.vote != 0 : true;
        be.refill = (uint32_t)(be.opt.refill > 0 ? be.opt.refill : (be.opt.refill == 0 ? 64 : ((be.vote && be.feat_trace == FEAT_SIMPLE) ? 32 : 16)));
        be.geom4 = be.sc.n_nodes4 ? 0xffffffffu : LN_V4 * be.sc.n_nodes2 + 9u * be.sc.n_prims;
        const uint32_t chunks = (n + 63u) / 64u, gmax = (uint32_t)scene->n_cu * 8u * (uint32_t)(be.opt.grid_mult ? be.opt.grid_mult : 8); // (one launch at a time: the single-lane segmentation)
        const uint32_t G = chunks < gmax ? chunks : gmax, seg_cap = ((chunks + G - 1) / G) * 64u;
        if ((uint64_t)G * seg_cap * 16ull >= (1ull << 32)) { g_err = "ptrs_trace_bench: too many rays for one launch (queue positions are 32-bit byte offsets of 16-byte records: fewer than 2^28)"; return PTRS_ERR_INVALID; }
        be.G = G; be.seg_cap = seg_cap;
        // the rays in queue order, as a frame's shade stage leaves them: ray i = 64 c + k sits at position (c mod G) seg_cap + (c / G) 64 + k, or
        // -- option `deal` -- in runs of whole segments ((c / cps) seg_cap + (c mod cps) 64 + k): a dump of a frame
        // dealt by image region comes segment by segment, and so keeps its regions
        const uint32_t cps = seg_cap / 64u; const bool by_region = be.deal_by_region() != 0;
        auto seg_of = [&](uint32_t c) { return by_region ? c / cps : c % G; };
        auto pos_of = [&](uint32_t c) { return (size_t)seg_of(c) * seg_cap + (size_t)(by_region ? c % cps : c / G) * 64u; };
        const size_t nq = (size_t)G * seg_cap;
        std::vector<v4> ro(nq), rd(nq);
        std::vector<uint32_t> q(nq, 0u), cnt((size_t)Q_STRIDE * G, 0u);
        for (uint32_t c = 0; c < chunks; ++c) {
            const uint32_t s = seg_of(c), m = c * 64u + 64u <= n ? 64u : n - c * 64u;
            for (uint32_t k = 0; k < m; ++k) {
                const uint32_t i = c * 64u + k; const size_t e = pos_of(c) + k;
                q[e] = i;
                ro[e].x = rays[7 * i]; ro[e].y = rays[7 * i + 1]; ro[e].z = rays[7 * i + 2]; ro[e].w = rays[7 * i + 6]; rd[e].x = rays[7 * i + 3]; rd[e].y = rays[7 * i + 4]; rd[e].z = rays[7 * i + 5]; rd[e].w = 0.0f;
            }
            cnt[(size_t)Q_EXT * G + s] += m;
        }
        DevBuf bo, bd, bh, bq, bc, bt, bs;
        auto cleanup = [&]() { bo.release(); bd.release(); bh.release(); bq.release(); bc.release(); bt.release(); bs.release(); };
        int rc;
        const size_t tk_words = (size_t)(repeats + 1) * TK_LAUNCH_WORDS + 4;
        if ((rc = upload(bo, ro)) || (rc = upload(bd, rd)) || (rc = bh.ensure(nq * 16)) || (rc = upload(bq, q)) || (rc = upload(bc, cnt)) || (rc = bt.ensure(tk_words * 4)) || (rc = bs.ensure(CNT_NUM * 8))) { cleanup(); return rc; }
        hipError_t e = hipMemset(bt.p, 0, tk_words * 4);
        if (e == hipSuccess) e = hipMemset(bs.p, 0, CNT_NUM * 8);
        DPaths P; std::memset(&P, 0, sizeof(P)); P.ray_o[0] = (v4 *)bo.p; P.ray_d[0] = (v4 *)bd.p; P.hit = (u4 *)bh.p;
        DQueues Q; std::memset(&Q, 0, sizeof(Q)); Q.ext[0] = (uint32_t *)bq.p; Q.counts = (uint32_t *)bc.p; Q.stats = (unsigned long long *)bs.p; Q.tickets = (uint32_t *)bt.p; Q.alive = (uint32_t *)bt.p + (size_t)(repeats + 1) * TK_LAUNCH_WORDS;
        DParams R; std::memset(&R, 0, sizeof(R)); R.n_paths = n;
        StackSpill sp = scene->spill;
        const HipBackend::TravFn fn = be.feat_trace == FEAT_FULL ? be.pick_extend<FEAT_FULL>(be.vote, sp.p != nullptr) : (be.feat_trace == FEAT_IMG_ENV ? be.pick_extend<FEAT_IMG_ENV>(be.vote, sp.p != nullptr) : be.pick_extend<FEAT_SIMPLE>(be.vote, sp.p != nullptr));
        const uint32_t grid = be.persistent_grid(fn, HipBackend::T_EXTEND);
        std::vector<hipEvent_t> ev(2 * (size_t)repeats, nullptr);
        for (auto &x : ev) if (e == hipSuccess) e = hipEventCreate(&x); // (destroyed below on every path: nothing between here and there returns)
        for (uint32_t k = 0; k <= repeats && e == hipSuccess; ++k) { // launch 0 counts nodes and triangles (untimed), 1 .. repeats are timed
            R.counters_on = k == 0 ? 1u : 0u;
            if (k) (void)hipEventRecord(ev[2 * (k - 1)], nullptr);
            hipLaunchKernelGGL(fn, dim3(grid), dim3(BLOCK), 0, nullptr, R, be.sc, sp, P, Q, 0u, seg_cap, be.refill, 0u, G, (uint32_t *)bt.p + (size_t)k * TK_LAUNCH_WORDS);
            if (k) (void)hipEventRecord(ev[2 * (k - 1) + 1], nullptr);
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipGetLastError();
        double ms = 0.0;
        for (uint32_t k = 0; k < repeats && e == hipSuccess; ++k) { float t = 0.0f; e = hipEventElapsedTime(&t, ev[2 * k], ev[2 * k + 1]); ms += t; }
        for (auto &x : ev) if (x) (void)hipEventDestroy(x);
        if (e == hipSuccess) {
            unsigned long long hs[CNT_NUM];
            e = hipMemcpy(hs, bs.p, sizeof(hs), hipMemcpyDeviceToHost);
            stats->nodes_visited = hs[CNT_NODES]; stats->tris_tested = hs[CNT_TRIS];
        }
        if (e == hipSuccess && hits_out) {
            std::vector<u4> hh(nq);
            e = hipMemcpy(hh.data(), bh.p, nq * 16, hipMemcpyDeviceToHost);
            for (uint32_t c = 0; c < chunks; ++c) {
                const uint32_t m = c * 64u + 64u <= n ? 64u : n - c * 64u;
                for (uint32_t k = 0; k < m; ++k) {
                    const uint32_t i = c * 64u + k; const u4 h = hh[pos_of(c) + k];
                    hits_out[i].prim = hit_prim(h.x); hits_out[i].b0 = u2f(h.y); hits_out[i].b1 = u2f(h.z); hits_out[i].b2 = u2f(h.w); hits_out[i].t = 0.0f;
                }
            }
        }
        stats->ms_trace = ms; stats->ms_extend = ms; stats->trace_launches = repeats; stats->extend_launches = repeats; stats->kernel_launches = repeats + 1; stats->rays_extension = (uint64_t)n * repeats;
        stats->queue_segments = G; stats->grid_wgs[0] = grid; stats->resident_wgs_per_cu[0] = be.last_per_cu[HipBackend::T_EXTEND];
        cleanup();
        if (e != hipSuccess) { g_err = std::string("ptrs_trace_bench: ") + hipGetErrorString(e); return PTRS_ERR_DEVICE; }
        return PTRS_OK;
    });
}

int ptrs_selftest_div3(int32_t device, uint32_t mode, uint64_t n_sets, uint64_t seed, uint64_t *mismatches_out, uint64_t *fast_path_sets_out, uint32_t *first_bad_out) {
    return guarded([&]() -> int {
        if (!mismatches_out || mode > 4u || n_sets == 0) { g_err = "bad self-test request"; return PTRS_ERR_INVALID; }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (this library has no CPU fallback)"; return PTRS_ERR_DEVICE; }
        if (device < 0 || device >= ndev) { g_err = "device ordinal out of range"; return PTRS_ERR_INVALID; }
        HIPCHK(hipSetDevice(device));
        DevBuf out, fb;
        int rc;
        if ((rc = out.ensure(16)) != PTRS_OK || (rc = fb.ensure(44)) != PTRS_OK) { out.release(); fb.release(); return rc; }
        hipError_t e = hipMemset(out.p, 0, 16);
        if (e == hipSuccess) e = hipMemset(fb.p, 0, 44);
        const uint32_t grid = 256u * 16u;
        const uint64_t threads = (uint64_t)grid * BLOCK, per_thread = (n_sets + threads - 1) / threads;
        if (e == hipSuccess) { hipLaunchKernelGGL(k_selftest_div3, dim3(grid), dim3(BLOCK), 0, nullptr, seed, mode, per_thread, (unsigned long long *)out.p, (uint32_t *)fb.p); e = hipDeviceSynchronize(); }
        if (e == hipSuccess) e = hipGetLastError();
        unsigned long long h[2] = {0, 0}; uint32_t hb[11] = {0};
        if (e == hipSuccess) e = hipMemcpy(h, out.p, 16, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(hb, fb.p, 44, hipMemcpyDeviceToHost);
        out.release(); fb.release();
        if (e != hipSuccess) { g_err = std::string("ptrs_selftest_div3: ") + hipGetErrorString(e); return PTRS_ERR_DEVICE; }
        *mismatches_out = h[0];
        if (fast_path_sets_out) *fast_path_sets_out = h[1];
        if (first_bad_out) std::memcpy(first_bad_out, hb, 40);
        return PTRS_OK;
    });
}

int ptrs_sobol_samples(const PtrsRenderParams *params, uint32_t n, const int32_t *px, const int32_t *py, const uint64_t *sample_nums, const uint32_t *dims, float *out, uint64_t *index_out) {
    if (!params || (n && (!px || !py || !sample_nums || !dims || !out))) { g_err = "null argument"; return PTRS_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_err = "no HIP device available (this library has no CPU fallback)"; return PTRS_ERR_DEVICE; }
    HIPCHK(hipSetDevice(params->device));
    if (n == 0) return PTRS_OK;
    const SampleGrid g = make_sample_grid(params->width, params->height, params->spp);
    for (uint32_t i = 0; i < n; ++i) {
        if (dims[i] >= 1024u) { g_err = "sobol sampler can only sample up to 1024 dimensions"; return PTRS_ERR_INVALID; }
        if (px[i] < g.min_x || px[i] >= g.min_x + g.NX || py[i] < g.min_y || py[i] >= g.min_y + g.NY) { g_err = "pixel outside the sample bounds"; return PTRS_ERR_INVALID; }
    }
    SobolDevice *sob;
    int rc = get_sobol(params->device, &sob);
    if (rc != PTRS_OK) return rc;
    DSampler S;
    S.matrices = (const uint32_t *)sob->matrices.p; S.bytetab = (const uint32_t *)sob->bytetab.p; S.nibtab = (const uint32_t *)sob->nibtab.p; S.vdc = (const uint64_t *)sob->vdc.p + (size_t)(g.log2_res - 1) * sob->stride; S.vdc_inv = (const uint64_t *)sob->vdc_inv.p + (size_t)(g.log2_res - 1) * sob->stride;
    S.log2_res = g.log2_res; S.resolution = g.resolution; S.min_x = g.min_x; S.min_y = g.min_y; S.spp = g.spp;
    DevBuf bx, by, bn, bdm, bo, bi;
    auto cleanup = [&]() { bx.release(); by.release(); bn.release(); bdm.release(); bo.release(); bi.release(); };
    if ((rc = bx.ensure((size_t)n * 4)) || (rc = by.ensure((size_t)n * 4)) || (rc = bn.ensure((size_t)n * 8)) || (rc = bdm.ensure((size_t)n * 4)) || (rc = bo.ensure((size_t)n * 4)) || (rc = bi.ensure((size_t)n * 8))) { cleanup(); return rc; }
    hipError_t e = hipMemcpy(bx.p, px, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(by.p, py, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(bn.p, sample_nums, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(bdm.p, dims, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sobol, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, nullptr, S, n, (const int32_t *)bx.p, (const int32_t *)by.p, (const uint64_t *)bn.p, (const uint32_t *)bdm.p, (float *)bo.p, (uint64_t *)bi.p);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = hipMemcpy(out, bo.p, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && index_out) e = hipMemcpy(index_out, bi.p, (size_t)n * 8, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) { g_err = std::string("ptrs_sobol_samples: ") + hipGetErrorString(e); return PTRS_ERR_DEVICE; }
    return PTRS_OK;
}

} // extern "C"
