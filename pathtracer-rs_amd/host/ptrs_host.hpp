// ptrs_host.hpp -- C++ host mirror of the reference's API around the render() hot path.
//
// The reference is compiled code (Rust); no Rust toolchain exists in the build image, so the host
// side above the C ABI (include/ptrs.h) is mirrored in C++ with the reference's names and
// argument meaning:
//   importer::import(path, resolution, default_lights)  src/common/importer/mod.rs:6-25 (.xml and .gltf/.glb branches)
//   from_gltf / RenderScene::from_gltf                  src/common/importer/gltf.rs, src/pathtracer/importer/gltf.rs (ptrs_gltf.cpp)
//   Camera::new / get_camera                            src/common/mod.rs:33-62, importer/mitsuba.rs:685-710
//   Film::{new, clear, get_sample_bounds, to_rgba_image} src/common/film.rs:132-251
//   RenderScene::from_mitsuba                           src/pathtracer/importer/mitsuba.rs:84-428 (rgb-parameter subset)
//   SamplerBuilder::new                                 src/pathtracer/sampler/sobol.rs:35-60
//   PathIntegrator::{new, preprocess, render}           src/pathtracer/integrator.rs:230,250,536
// The Python package (pathtracer-rs_amd/*.py) mirrors the same surface for tests and bench.py; both
// produce bit-identical scene descriptions (tests/test_host_cpp.py).
#pragma once
#include <cstdint>
#include <memory>
#include <functional>
#include <string>
#include <vector>

#include "../../include/ptrs.h"

namespace ptrs_host {

struct Film {
    int width = 0, height = 0;
    std::vector<PtrsFilmPixel> pixels; // row-major, accumulated (film.rs:113-129)
    Film() = default;
    Film(int w, int h) : width(w), height(h), pixels((size_t)w * h, PtrsFilmPixel{{0, 0, 0}, 0}) {}
    void clear();                                   // film.rs:164-172
    void get_sample_bounds(int32_t out[4]) const;   // film.rs:174-185: min_x, min_y, max_x, max_y
    std::vector<uint8_t> to_rgba_image() const;     // film.rs:230-251 + spectrum.rs:95-102 (sRGB, +0.5, clamp)
};

struct Camera {
    PtrsCamera abi{};
    Film film;
};

struct Mesh {
    std::vector<float> pos, normal, uv, tangent;
    std::vector<uint32_t> indices;
    int32_t material = -1;
    int32_t alpha_mask_tex = -1;
};

// MIP pyramid of an image texture (MIPMap::new, texture.rs:279-405), owned by the scene
struct TexImage {
    int channels = 0;
    std::vector<std::vector<float>> levels; // level l: rows*cols*channels, row-major
    std::vector<int32_t> cols, rows;
    std::vector<const float *> ptrs;
};
// Distribution2D of an infinite light (sampling.rs:185-230)
struct EnvDistribution {
    int nu = 0, nv = 0;
    std::vector<float> func, cdf, func_int, marg_cdf;
    float marg_func_int = 0.0f;
};

struct RenderScene {
    std::vector<Mesh> meshes;
    std::vector<PtrsMaterial> materials;
    std::vector<PtrsTexture> textures;
    std::vector<PtrsLight> lights;
    std::vector<std::unique_ptr<TexImage>> images;          // referenced by textures[i].level_data
    std::vector<std::unique_ptr<EnvDistribution>> env_dists; // referenced by lights[i].dist_*
    // flat description for ptrs_scene_create (valid while *this is alive and unmodified)
    const PtrsSceneDesc &desc();
    size_t num_triangles() const;

private:
    std::vector<PtrsMesh> abi_meshes_;
    PtrsSceneDesc desc_{};
};

struct SamplerBuilder { // sobol.rs:25-60
    int samples_per_pixel = 1;
    int32_t sample_bounds[4] = {0, 0, 0, 0};
    int resolution = 1;
    uint32_t log_2_resolution = 0;
    SamplerBuilder() = default;
    SamplerBuilder(int spp, const int32_t bounds[4]);
};

class PathIntegrator { // integrator.rs:219-246
public:
    PathIntegrator(const SamplerBuilder &sb, int max_depth, bool show_progress_bar = false, int device = 0);
    ~PathIntegrator();
    void preprocess(const RenderScene &scene);          // integrator.rs:250-258
    void toggle_progress_bar() { show_progress_bar_ = !show_progress_bar_; }
    // integrator.rs:536-642 on the GPU; accumulates into camera.film.  Returns PTRS_OK or an error code.
    int render(Camera &camera, RenderScene &scene, PtrsStats *stats = nullptr);
    // When set, render() publishes the film after every pass of the pipeline (ptrs_render_progressive) and calls this with
    // (passes done, passes in total, first row, one past the last row that changed): the hook a preview gets instead of the
    // reference's second thread that reads the film every 2 s (headless.rs:197-214).
    std::function<void(uint32_t, uint32_t, int32_t, int32_t)> on_pass;
    // The same render split over the devices 0 .. n_devices-1 of this process: the scene is uploaded to every
    // device, output rows are cut into one band per device -- weighted by the ray counts of a 1-spp probe when
    // cost_weighted and the plan takes at least 1 % off the slowest band; planned once per view -- and ptrs_render_multi renders the bands on one host thread per device and gathers them.  The film
    // is bit-identical to render()'s.  stats: n_devices records or nullptr; bands_out: n_devices + 1 row bounds.
    int render_multi(Camera &camera, RenderScene &scene, int n_devices, bool cost_weighted = true, PtrsStats *stats = nullptr, std::vector<int32_t> *bands_out = nullptr);
    std::string last_error;

private:
    SamplerBuilder sb_;
    int max_depth_;
    float rr_threshold_ = 1.0f;
    int rr_start_depth_ = 3;
    bool rr_enable_ = true;
    bool show_progress_bar_;
    int device_;
    PtrsScene *gpu_scene_ = nullptr;
    const RenderScene *gpu_scene_src_ = nullptr;
    std::vector<PtrsScene *> multi_scenes_; // one per device, device i at index i
    const RenderScene *multi_src_ = nullptr;
    std::string plan_key_;          // the view (camera, resolution, depth, device count) plan_ was made for
    std::vector<int32_t> plan_;     // its row bounds
    PtrsRenderParams params(const Camera &camera) const;
};

// Per-row cost of a frame for band planning: the BVH queries of every sample row's paths in ONE 1-spp render with device counters
// (ptrs_render_row_cost: a few milliseconds).  Returns false and fills err on failure.
bool probe_row_cost(PtrsScene *scene, const PtrsCamera &camera, const PtrsRenderParams &params, int strips, std::vector<float> &row_cost, std::string &err);

// importer::import for Mitsuba XML (rectangle / cube shapes, twosided / diffuse / conductor /
// roughconductor / dielectric / plastic / roughplastic bsdfs with rgb parameters, area emitters,
// perspective sensor).  Returns false and fills err on failure.
bool import_scene(const std::string &path, int res_w, int res_h, Camera &camera, RenderScene &scene, std::string &err,
                  bool default_lights = false, const std::string &env_map_path = "");
// the .gltf / .glb branch (ptrs_gltf.cpp).  default_lights adds the environment light of `--default_lights`; the
// reference reads data/abandoned_tank_farm_04_1k.hdr from its source tree, here the Radiance file is named by the caller.
bool import_gltf(const std::string &path, int res_w, int res_h, bool default_lights, const std::string &env_map_path,
                 Camera &camera, RenderScene &scene, std::string &err);
// Camera::new (common/mod.rs:33-62) from an isometry (unit quaternion i,j,k,w + translation) and Perspective3::new arguments
void make_camera_perspective(const float rot_ijkw[4], const float trans[3], float aspect, float fovy, float znear, float zfar, int res_w, int res_h, Camera &cam);

bool write_png_rgba8(const std::string &path, int w, int h, const std::vector<uint8_t> &rgba, std::string &err);
bool dump_scene(const std::string &path, const Camera &camera, const RenderScene &scene);
bool dump_scene_full(const std::string &path, const Camera &camera, const RenderScene &scene); // every field of the flat description

} // namespace ptrs_host
