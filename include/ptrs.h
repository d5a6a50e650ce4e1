/*
 * ptrs.h -- C ABI of the MI355X wavefront path-tracing backend ("ptrs").
 *
 * Drop-in boundary: the reference has no FFI; its hot path is entered through one Rust method
 *     impl PathIntegrator { pub fn render(&self, camera: &Camera, scene: &RenderScene) }
 *                                                   (src/pathtracer/integrator.rs:536)
 * called from src/headless.rs:216,227, src/viewer/mod.rs:115, benches/benchmark_pathtracer.rs:30.
 * Everything that call reads (immutable scene, camera, sampler/integrator parameters) crosses
 * this ABI as flat plain-old-data; the only thing it writes (the film accumulators,
 * src/common/film.rs:113-129) comes back as 16-byte pixels.  INTEGRATION.md shows the Rust
 * `extern "C"` binding a maintainer would add.
 *
 * Conventions: caller owns every host buffer (they may be freed after ptrs_scene_create /
 * ptrs_render return); the library owns device memory; all entry points return PTRS_OK (0) or a
 * negative error code and never abort; ptrs_last_error() gives the message for the calling
 * thread.  One render at a time per PtrsScene.  All matrices are row-major 4x4.
 */
#ifndef PTRS_H
#define PTRS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTRS_ABI_VERSION 4

enum {
    PTRS_OK = 0,
    PTRS_ERR_INVALID = -1,     /* bad argument / inconsistent description */
    PTRS_ERR_UNSUPPORTED = -2, /* a record kind this build does not implement */
    PTRS_ERR_DEVICE = -3,      /* HIP runtime failure (no GPU, OOM, launch error) */
    PTRS_ERR_IO = -4
};

/* ---- textures (src/pathtracer/texture.rs:15-192,238-465) ---------------------------------- */
enum { PTRS_TEX_CONSTANT = 0, PTRS_TEX_CHECKER = 1, PTRS_TEX_IMAGE = 2 };
enum { PTRS_WRAP_REPEAT = 0, PTRS_WRAP_BLACK = 1, PTRS_WRAP_CLAMP = 2 }; /* common/mod.rs:65-70 */

typedef struct PtrsTexture {
    int32_t kind;     /* PTRS_TEX_* */
    int32_t channels; /* 1 = f32 texture, 3 = Spectrum / Vector3 texture */
    float value[3];   /* CONSTANT: the value.  CHECKER: v1 (texture.rs:56-60) */
    float value2[3];  /* CHECKER: v2 */
    float su, sv, du, dv; /* UVMap (texture.rs:29-53); ignored by CONSTANT */
    int32_t wrap;     /* IMAGE: PTRS_WRAP_* */
    int32_t n_levels; /* IMAGE: MIP pyramid built by the host exactly as MIPMap::new (279-405) */
    const float *const *level_data; /* n_levels pointers; level l is rows*cols*channels f32,
                                        row-major, row index = t, column = s (texel(): 245-273) */
    const int32_t *level_cols;
    const int32_t *level_rows;
} PtrsTexture;

/* ---- materials (src/pathtracer/material/{mod,metal,substrate,disney}.rs) ------------------- */
enum {
    PTRS_MAT_MATTE = 0,     /* tex[0]=kd                                   mod.rs:143-167 */
    PTRS_MAT_METAL = 1,     /* tex[0]=eta [1]=k [2]=r [3]=roughness [4]=u_rough [5]=v_rough
                               (-1 = absent), flags&1 = remap_roughness    metal.rs:49-94 */
    PTRS_MAT_MIRROR = 2,    /*                                             mod.rs:169-195 */
    PTRS_MAT_GLASS = 3,     /* tex[0]=kr [1]=kt [2]=index                  mod.rs:197-256 */
    PTRS_MAT_DISNEY = 4,    /* tex[0]=color [1]=metallic [2]=eta [3]=roughness  disney.rs:172-264 */
    PTRS_MAT_SUBSTRATE = 5, /* tex[0]=kd [1]=ks [2]=nu [3]=nv, flags&1 = remap  substrate.rs:42-68 */
    PTRS_MAT_NORMAL = 6     /* tex[0]=normal map (3-channel), inner = wrapped material  mod.rs:39-79,125-141 */
};

typedef struct PtrsMaterial {
    int32_t kind;
    int32_t tex[6];
    int32_t flags;
    int32_t inner;
} PtrsMaterial;

/* ---- geometry (src/pathtracer/shape.rs:581-641, primitive.rs:20-24) ------------------------ */
typedef struct PtrsMesh {
    uint32_t n_verts;
    uint32_t n_tris;
    const float *pos;        /* n_verts*3, WORLD space (TriangleMesh::new_with_transform 592-623) */
    const float *normal;     /* n_verts*3 or NULL */
    const float *tangent;    /* n_verts*3 or NULL ("s") */
    const float *uv;         /* n_verts*2 or NULL -> default (0,0),(1,0),(1,1) (shape.rs:34-48) */
    const uint32_t *indices; /* n_tris*3 */
    int32_t material;        /* index into materials[] (one GeometricPrimitive per triangle) */
    int32_t alpha_mask_tex;  /* 1-channel texture or -1 (shape.rs:228-244,471-521) */
    int32_t reverse_orientation;        /* always 0 in the reference (shape.rs:626-641) */
    int32_t transform_swaps_handedness; /* always 0 in the reference */
} PtrsMesh;

/* ---- lights (src/pathtracer/light.rs) ------------------------------------------------------ */
enum { PTRS_LIGHT_POINT = 0, PTRS_LIGHT_DIRECTIONAL = 1, PTRS_LIGHT_AREA = 2, PTRS_LIGHT_INFINITE = 3 };

typedef struct PtrsLight {
    int32_t kind;
    float v[3]; /* POINT: p_light (86-97).  DIRECTIONAL: w_light, already normalised (159-171) */
    float c[3]; /* POINT: I.  DIRECTIONAL: L */
    /* AREA: one DiffuseAreaLight per emissive triangle (231-250; mitsuba.rs:309-323) */
    uint32_t mesh, tri;
    int32_t ke_tex; /* 3-channel emission texture */
    /* DIRECTIONAL / INFINITE: Light::preprocess (209-211, 480-482; bounds.rs:126-134) */
    float world_center[3];
    float world_radius;
    /* INFINITE (321-503) */
    int32_t lmap_tex; /* 3-channel IMAGE texture (the MIPMap), wrap = REPEAT */
    float light_to_world[16];
    float world_to_light[16];
    int32_t dist_nu, dist_nv;   /* Distribution2D dimensions (sampling.rs:185-230) */
    const float *dist_func;     /* nv*nu      conditional func */
    const float *dist_cdf;      /* nv*(nu+1)  conditional cdf */
    const float *dist_func_int; /* nv         conditional integrals (= marginal func) */
    const float *marg_cdf;      /* nv+1 */
    float marg_func_int;
} PtrsLight;

/* Optional pre-built accelerator in the reference's flattened layout (accelerator.rs:89-95).
 * When nodes == NULL the library builds its own BVH (closest-hit results do not depend on the
 * tree: SURVEY.md Q29). */
typedef struct PtrsBvhNode {
    float p_min[3];
    float p_max[3];
    uint32_t offset;    /* leaf: first primitive in prim order; interior: second child */
    uint16_t num_prims; /* 0 = interior */
    uint8_t axis;
    uint8_t pad;
} PtrsBvhNode;

typedef struct PtrsSceneDesc {
    uint32_t n_meshes;
    const PtrsMesh *meshes;
    uint32_t n_materials;
    const PtrsMaterial *materials;
    uint32_t n_textures;
    const PtrsTexture *textures;
    uint32_t n_lights;
    const PtrsLight *lights; /* order = RenderScene::lights (sampled uniformly, integrator.rs:204) */
    uint32_t n_bvh_nodes;    /* optional */
    const PtrsBvhNode *bvh_nodes;
    const uint32_t *bvh_prims; /* n_prims entries: global triangle id (mesh-major) in leaf order */
} PtrsSceneDesc;

/* ---- camera (src/common/mod.rs:20-62; pathtracer/mod.rs:59-81) ----------------------------- */
typedef struct PtrsCamera {
    float rot[4];   /* cam_to_world rotation, unit quaternion (i, j, k, w) = nalgebra Isometry3 */
    float trans[3]; /* cam_to_world translation */
    float m00, m11, m22, m23;   /* Perspective3 matrix entries used by unproject_point */
    float raster_to_screen[16]; /* Affine3 */
    float dx_camera[3];
    float dy_camera[3];
} PtrsCamera;

/* ---- render parameters (sampler/sobol.rs:35-60; integrator.rs:219-246) --------------------- */
typedef struct PtrsRenderParams {
    int32_t width, height; /* film resolution (film.rs:132) */
    int32_t spp;           /* rounded up to a power of two like the reference (sobol.rs:37) */
    int32_t max_depth;
    float rr_threshold;     /* reference: 1.0 */
    int32_t rr_start_depth; /* reference: 3 */
    int32_t rr_enable;      /* reference: 1 */
    int32_t row_begin, row_end; /* output rows written by this call; 0,height = whole film.
                                   Multi-GPU ranks each take a band */
    int32_t device;             /* HIP device ordinal */
    uint32_t paths_per_pass;    /* 0 = auto */
    uint32_t flags;             /* PTRS_FLAG_* */
    int32_t sampler;            /* PTRS_SAMPLER_*: the reference builds a SobolSamplerBuilder (main.rs:103); its StratifiedSampler
                                   (sampler/stratified.rs) is compiled but never instantiated (sampler/mod.rs:169-170) */
    int32_t n_sampled_dimensions; /* STRATIFIED: StratifiedSamplerBuilder::new's n_sampled_dimensions; spp = dim_pixel_samples^2.
                                   Must cover the path: >= 3 * (max_depth + 1) + 1, because a draw past it would come straight
                                   from the tile's generator in path order (mod.rs:137-151), which a wavefront cannot reproduce:
                                   such a render is refused (PTRS_ERR_UNSUPPORTED) */
} PtrsRenderParams;
enum { PTRS_SAMPLER_SOBOL = 0, PTRS_SAMPLER_STRATIFIED = 1 };

enum {
    PTRS_FLAG_COUNTERS = 1u, /* fill nodes_visited / tris_tested (slower: device atomics) */
    PTRS_FLAG_TIMING = 2u,   /* hipEvent timing of every kernel launch (needs stream syncs) */
    PTRS_FLAG_FILM_ZERO = 4u /* ptrs_render_multi: film_inout is all zeros (Film::clear), the devices clear their bands instead of uploading them */
};

/* film pixel: linear RGB sums and filter-weight sum (film.rs:113-119; splat_xyz is never
 * written by the reference and is not carried).  Row-major, y*width + x (film.rs:187-191). */
typedef struct PtrsFilmPixel {
    float rgb[3];
    float weight;
} PtrsFilmPixel;

typedef struct PtrsStats {
    uint64_t samples;        /* li() evaluations = (W+4)(H_band+4)*spp */
    uint64_t rays_extension; /* BVH closest-hit queries, integrator.rs:416 */
    uint64_t rays_shadow;    /* BVH any-hit queries, light.rs:40 */
    uint64_t rays_mis;       /* BVH closest-hit queries, integrator.rs:119 */
    uint64_t nodes_visited;  /* only with PTRS_FLAG_COUNTERS */
    uint64_t tris_tested;    /* only with PTRS_FLAG_COUNTERS */
    uint64_t passes;
    uint64_t kernel_launches;
    uint64_t trace_launches;
    double ms_total;    /* wall clock of the call (host timer around the stream) */
    double ms_trace;    /* sum of trace-kernel durations (PTRS_FLAG_TIMING) */
    double ms_shade;    /* generate + sort + shade + resolve kernels (PTRS_FLAG_TIMING) */
    double ms_film;     /* film kernels (PTRS_FLAG_TIMING) */
    uint64_t bvh_nodes;
    uint64_t bvh_max_depth;
    uint64_t device_bytes; /* peak device allocation of the call */
    /* PTRS_FLAG_TIMING, per kernel class: durations summed over the launches of the call, and the launch counts.
     * ms_trace = ms_extend + ms_connect; ms_shade = ms_shade_kernels + ms_aux. */
    double ms_extend;        /* extension-ray traversal kernels (k_extend / k_extend_rf) */
    double ms_connect;       /* shadow + MIS ray traversal kernels (k_connect / k_connect_rf) */
    double ms_shade_kernels; /* k_shade<material, features> */
    double ms_aux;           /* k_generate, k_epilogue, k_resolve */
    uint64_t extend_launches, connect_launches, shade_launches, aux_launches, film_launches;
    uint64_t error_flags;    /* PTRS_ERRFLAG_* raised by device code (render returns PTRS_ERR_UNSUPPORTED) */
    /* PTRS_FLAG_COUNTERS, lane-refill traversal kernels: lane occupancy of the two phases =
     * node_visits / node_steps_x64 and tris_tested (of these kernels) / tri_steps_x64 */
    uint64_t node_steps_x64, node_visits, tri_steps_x64;
    uint64_t debug[12];      /* zero in product builds; diagnostic builds (-DPTRS_STAMPS): wave-clock sums per phase of k_shade */
    /* how the queue kernels of the call's last pass were launched (ABI 3): segments per queue (one wave each), and per kernel class
     * [0] extend, [1] connect, [2] shade, [3] aux the workgroups of the last launch and the resident workgroups per CU the launch was
     * sized for (hipOccupancyMaxActiveBlocksPerMultiprocessor) */
    uint64_t queue_segments;
    uint64_t grid_wgs[4];
    uint64_t resident_wgs_per_cu[4];
    uint64_t lanes;          /* pipeline lanes the call ran on (option "lanes", or chosen by the size of the job) */
    uint64_t grid_pct;       /* share of a kernel's resident capacity its launches took (option "grid_pct", or chosen with the lanes) */
    double ms_enqueue;       /* host time from the start of the call until its last pass was enqueued (a job of more passes than lanes
                              * waits for a lane in between); ms_total - ms_enqueue is what the host then waited for the device */
    /* ABI 4: the fused tail (one launch per pass in which every wave takes its queue segment through all remaining rounds) */
    double ms_tail;          /* its kernels (PTRS_FLAG_TIMING; their traversal and shading are not in ms_trace / ms_shade) */
    uint64_t tail_launches;
    uint64_t tail_round;     /* the round at which the call's last pass handed over, 0xffffffff: it did not */
} PtrsStats;

enum {
    PTRS_ERRFLAG_SOBOL_DIM = 1u,  /* a path needed Sobol dimension >= 1024 (the reference panics: sobol.rs:177-183) */
    PTRS_ERRFLAG_NULL_SKIPS = 2u  /* paths still alive after max_depth + 1 + 64 rounds of null-BSDF skips */
};

typedef struct PtrsScene PtrsScene;

/* Process-wide tuning knobs; the library reads no environment variables.  Names: "lanes" (1-8 concurrent pipeline lanes;
 * 0 = four, one for a job under 4 M paths), "grid_mult" (queue segments -- one wave each -- per pass = CUs x 8 x grid_mult; 0 = 1
 * with several lanes -- up to 4 for scenes whose tree is traversed out of HBM / L2, by the paths of a pass --, 8 with one), "grid_pct" (share of its resident capacity a persistent launch takes; 0 = 100, or 50 with several
 * lanes and a grid_mult given by hand), "persist" (0/1), "whole_rounds" (0/1), "refill" / "refill_connect" (idle-lane threshold of the
 * lane-refill traversal kernels, 0 = refill only when the whole wave is idle, -1 = by scene), "vote" (phase voting in the traversal
 * kernels: 0 off, 1 on, 2 extension kernel only, -1 = by scene), "stack_lds" (8 or 16 LDS stack entries per lane), "shade_lds"
 * (0/1: shade kernels read their small tables from LDS), "env_presample" (0/1: environment-light samples evaluated ahead of the
 * shade kernels), "fused_epilogue" / "fused_resolve" (0/1: the traversal kernels run the segment's epilogue / MIS resolve behind
 * the wave's last ray instead of separate k_epilogue / k_resolve launches), "node_form" (0 auto, 2 force quad nodes), "node_order"
 * (0 / 1: quad-node order behind the LDS-cached top), "workspace_pct" (share of the free device memory the render workspace may
 * take, default 40), "peer_copy" (0: ptrs_render_multi stages bands through the host film).  None of them changes a result bit;
 * they select between equivalent schedules.  Scene-level knobs (node_form, node_order, stack_lds) are read by
 * ptrs_scene_create, the rest by each render call.  Replaces nothing in the reference (its only knobs are the CLI flags of
 * main.rs:36-52). */
int ptrs_set_option(const char *name, int64_t value);
int ptrs_get_option(const char *name, int64_t *value);
/* The same knobs for ONE scene: its renders take this value instead of the process-wide one (two embedders in a process, or
 * two scenes of one embedder, can differ).  Knobs read at scene creation (node_form, stack_lds) are not affected. */
int ptrs_scene_set_option(PtrsScene *scene, const char *name, int64_t value);

int ptrs_abi_version(void);
/* Identity of this build: a hash over the kernel sources and every compiler flag (pathtracer-rs_amd/build.py).  Measurement files
 * under profiles/ record it; bench.py takes hardware counters only from a file measured with the library it runs. */
const char *ptrs_build_id(void);
int ptrs_abi_sizeof(int which); /* sizeof the ABI structs as compiled (binding self-check) */
const char *ptrs_last_error(void);

/* Uploads the scene and (unless desc->bvh_nodes is given) builds the accelerator on the host.
 * Replaces: RenderScene construction hand-off, the precedent being OptixAccelerator::new(&scene)
 * (src/pathtracer/gpu/optix.rs:160-290) which consumes the same flat mesh arrays. */
int ptrs_scene_create(const PtrsSceneDesc *desc, int32_t device, PtrsScene **out);
void ptrs_scene_destroy(PtrsScene *scene);
/* accelerator facts (the reference logs these: accelerator.rs:131-148); any pointer may be NULL */
int ptrs_scene_info(PtrsScene *scene, uint64_t *n_nodes, uint64_t *max_depth, uint64_t *n_tris);

/* PathIntegrator::render (integrator.rs:536-642).  ACCUMULATES into film_inout (host memory,
 * width*height pixels) like Film::merge_film_tile (film.rs:213-228); rows outside
 * [row_begin,row_end) are untouched.  stats may be NULL. */
int ptrs_render(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params,
                PtrsFilmPixel *film_inout, PtrsStats *stats);

/* render with the film visible while it forms -- what the reference's preview thread gets by reading the shared film
 * every two seconds while render() runs (headless.rs:197-214).  After every pass of the
 * wavefront pipeline (a block of sample rows x a block of sample indices) whose film kernel has finished, the output rows
 * it touched are copied into film_inout and `fn(user, passes_done, passes_total, row_begin, row_end)` is called on the
 * calling thread; the rows hold the samples accumulated so far (rgb and weight sums: divide to display, film.rs:253-271):
 * at least those of the passes reported so far -- the copy is taken when the pass's pipeline lane is next waited for, passes
 * behind it keep running and may already show -- and callbacks arrive in pass order.
 * The final film is bit-identical to ptrs_render's. */
typedef void (*PtrsProgressFn)(void *user, uint32_t passes_done, uint32_t passes_total, int32_t row_begin, int32_t row_end);
int ptrs_render_progressive(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params,
                            PtrsFilmPixel *film_inout, PtrsProgressFn fn, void *user, PtrsStats *stats);

/* One process, several devices (the reference host is one process: main.rs:101-126).  scenes[i] is the same scene
 * created on device i (ptrs_scene_create with that ordinal); output rows [band_bounds[i], band_bounds[i+1]) are rendered
 * by scenes[i] on a host thread of its own, gathered into scenes[0]'s device with peer copies and returned in
 * film_inout (host, ACCUMULATED like ptrs_render).  band_bounds: n+1 row numbers from 0 to height, or NULL for equal
 * bands; ptrs_plan_bands computes them, optionally weighted by a per-row cost (row_cost[height], e.g. ray counts of
 * a 1-spp probe; NULL = equal rows).  stats_per_scene: n records or NULL.  Bit-identical to ptrs_render. */
int ptrs_plan_bands(int32_t height, uint32_t n, const float *row_cost, int32_t *bounds_out /* n + 1 */);
/* The per-row cost for ptrs_plan_bands, measured: `params`' render (give spp = 1) with one device counter per sample row -- every BVH
 * query a path makes is added to its row -- and no film; row_cost_out[height] = queries of the sample row under each output row.  One
 * call of a few milliseconds; what it replaces is the reference's dynamic tile queue (integrator.rs:617-637), which balances at run
 * time what a static band split has to know beforehand. */
int ptrs_render_row_cost(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, float *row_cost_out, PtrsStats *stats);
int ptrs_render_multi(PtrsScene *const *scenes, uint32_t n, const PtrsCamera *camera,
                      const PtrsRenderParams *params, const int32_t *band_bounds,
                      PtrsFilmPixel *film_inout, PtrsStats *stats_per_scene);

/* Same, but the film lives in DEVICE memory (width*height PtrsFilmPixel) and the work is queued
 * on `hip_stream` (a hipStream_t, NULL = default stream); returns after the stream has drained.
 * This is what bench.py times and what the multi-GPU gather reads. */
int ptrs_render_device(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params,
                       void *film_inout_device, void *hip_stream, PtrsStats *stats);

/* Test/debug form of render: additionally returns the radiance of every sample,
 * sample_rgb[((sy*(W+4) + sx)*spp + s)*3 + c] for sample-pixel (sx,sy) relative to the sample
 * bounds' p_min (film.rs:174-185), i.e. the value `l` at integrator.rs:579.  Host memory. */
int ptrs_render_samples(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params,
                        PtrsFilmPixel *film_inout, float *sample_rgb, PtrsStats *stats);

/* PathIntegrator::render_single_pixel (integrator.rs:505-534): radiance of every sample of one
 * pixel, rgb_out[spp*3]. */
int ptrs_render_single_pixel(PtrsScene *scene, const PtrsCamera *camera,
                             const PtrsRenderParams *params, int32_t px, int32_t py, float *rgb_out);

/* RenderScene::intersect / intersect_p (pathtracer/mod.rs:92-98 -> accelerator.rs:359-475) over a
 * batch of rays, run by the same traversal kernel the renderer uses.
 * rays: n * 7 floats (o.xyz, d.xyz, t_max).  any_hit = 0: closest hit, hits_out = n * PtrsHit.
 * any_hit = 1: hits_out[i].prim = 0 if occluded else -1. */
typedef struct PtrsHit {
    int32_t prim; /* global triangle id (mesh-major), -1 = miss */
    float t;
    float b0, b1, b2; /* barycentrics as computed by Triangle::intersect (shape.rs:157-160) */
} PtrsHit;
int ptrs_trace_rays(PtrsScene *scene, uint32_t n, const float *rays, int32_t any_hit,
                    PtrsHit *hits_out, PtrsStats *stats);

/* Traversal bench (the reference's precedent: benches/benchmark_pathtracer.rs:35-54, scene.intersect on one ray in a loop).
 * ptrs_render_dump_rays: starts `params`' render, stops in its first pass before round `round` and returns that round's extension
 * rays (round 0: the camera rays, round k: the rays leaving the paths' k-th vertices), at most max_rays records of 7 floats.
 * ptrs_trace_bench: closest-hit traversal of n rays with the frame's own extension kernel (lane refill, phase voting, persistent
 * waves), `repeats` timed launches after one counting launch: stats->ms_trace (sum of the timed launches), nodes_visited /
 * tris_tested (of ONE launch), rays_extension = n * repeats; hits_out (NULL or n records, t not filled) equals ptrs_trace_rays'. */
int ptrs_render_dump_rays(PtrsScene *scene, const PtrsCamera *camera, const PtrsRenderParams *params, uint32_t round,
                          uint32_t max_rays, float *rays_out, uint32_t *n_out);
int ptrs_trace_bench(PtrsScene *scene, uint32_t n, const float *rays, uint32_t repeats, PtrsHit *hits_out, PtrsStats *stats);

/* SobolSampler (sampler/sobol.rs): value of dimension dims[i] for sample sample_nums[i] of pixel
 * (px[i],py[i]) with the sampler built for `params` -- the device implementation of
 * start_pixel + get_index_for_sample + sample_dimension (81-114,169-193). */
int ptrs_sobol_samples(const PtrsRenderParams *params, uint32_t n, const int32_t *px,
                       const int32_t *py, const uint64_t *sample_nums, const uint32_t *dims,
                       float *out, uint64_t *index_out /* may be NULL */);

/* Self-test of the device code's shared-divisor division (csrc/pt_vec.h, div_shared3: one reciprocal for three quotients -- measured,
 * slower than the compiler's division in the shade kernels and therefore NOT on the render path; the reference divides component by
 * component, e.g. integrator.rs:63-75, 453-497) against the compiler's IEEE division, bit for bit, over
 * about n_sets generated operand sets (a0, a1, a2, b).  mode 0 random bit patterns, 1 exponent / mantissa edge cases, 2 render ranges,
 * 3 Russian-roulette ranges, 4 quotients next to 1 and to rounding ties.  first_bad_out (NULL or 10 words): a0 a1 a2 b, the three
 * quotients computed, the three expected. */
int ptrs_selftest_div3(int32_t device, uint32_t mode, uint64_t n_sets, uint64_t seed, uint64_t *mismatches_out,
                       uint64_t *fast_path_sets_out /* may be NULL */, uint32_t *first_bad_out /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* PTRS_H */
