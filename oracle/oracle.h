/* ORACLE -- TEST INFRASTRUCTURE ONLY.  C API of the CPU restatement of the reference's hot path
 * (see orc_math.h for the full notice).  Loaded by tests/ (ctypes) and by bench.py's
 * cpu_baseline leg; never by the product.  Parity unpinned. */
#ifndef ORACLE_H
#define ORACLE_H
#include "../include/ptrs.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct OrcScene OrcScene;
const char *orc_last_error(void);
int orc_load_tables(const char *path);
int orc_scene_create(const PtrsSceneDesc *desc, OrcScene **out);
void orc_scene_destroy(OrcScene *s);
int orc_scene_info(OrcScene *s, uint64_t *n_nodes, uint64_t *max_depth, uint64_t *n_tris);
int orc_scene_get_bvh(OrcScene *s, PtrsBvhNode *nodes_out, uint32_t *prims_out);
int orc_render(OrcScene *s, const PtrsCamera *cam, const PtrsRenderParams *p, PtrsFilmPixel *film_px,
               float *sample_rgb, int n_threads, PtrsStats *stats);
int orc_render_single_pixel(OrcScene *s, const PtrsCamera *cam, const PtrsRenderParams *p, int32_t px,
                            int32_t py, float *rgb_out);
int orc_trace_rays(OrcScene *s, uint32_t n, const float *rays, int32_t any_hit, int32_t brute_force,
                   PtrsHit *hits, PtrsStats *stats);
int orc_sobol_samples(const PtrsRenderParams *p, uint32_t n, const int32_t *px, const int32_t *py,
                      const uint64_t *sample_nums, const uint32_t *dims, float *out, uint64_t *index_out);
int orc_filter_table(float *out256);
uint32_t orc_log2_int(uint64_t v);
int orc_solve_2x2(const float *a, const float *b, float *x);
float orc_next_float_up(float v);
float orc_next_float_down(float v);
float orc_detmath(int fn, float x, float y);
int orc_stratified_tile(uint64_t seed, int32_t tile_w, int32_t tile_h, int32_t dim_pixel_samples, int32_t n_dims, float *out);
void orc_pcg64mcg(uint64_t state_lo, uint64_t state_hi, uint32_t n, uint64_t *out);
int orc_bsdf_eval(const PtrsMaterial *mat, const float *tex_values, uint32_t n, const float *wo,
                  const float *u, float *out);
#ifdef __cplusplus
}
#endif
#endif
