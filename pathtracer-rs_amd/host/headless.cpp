// ptrs_headless -- the reference's headless front-end over the HIP backend.
//
// Mirrors src/main.rs:35-145 (flag subset: SCENE, -o/--output DIR, -s/--samples, -r/--resolution WxH,
// -d/--max_depth, --default_lights, --headless; the viewer flags are accepted and ignored) and src/headless.rs:222-229
// (one-shot render, then film.to_rgba_image().save(DIR/render.png)).  Wiring follows main.rs:101-126:
//   import -> SamplerBuilder::new(spp, film.get_sample_bounds()) -> PathIntegrator::new(.., max_depth)
//   -> preprocess -> render -> save.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ptrs_host.hpp"

using namespace ptrs_host;

static bool parse_resolution(const char *s, int &w, int &h) { // main.rs:23-33
    const char *x = std::strchr(s, 'x');
    if (!x || std::strchr(x + 1, 'x')) return false;
    w = (int)std::strtof(std::string(s, x - s).c_str(), nullptr); h = (int)std::strtof(x + 1, nullptr);
    return w > 0 && h > 0;
}

int main(int argc, char **argv) {
    std::string scene_path, out_dir, dump_path, dump_full_path, env_map_path;
    bool default_lights = false, even_bands = false, preview = false, progress = false;
    int n_gpus = 1;
    int spp = 1, max_depth = 15, w = 640, h = 480; // DEFAULT_RESOLUTION common/mod.rs:14
    bool have_out = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "error: %s needs a value\n", name); std::exit(2); } return argv[++i]; };
        if (a == "-o" || a == "--output") { out_dir = need("--output"); have_out = true; }
        else if (a == "-s" || a == "--samples") spp = std::atoi(need("--samples"));
        else if (a == "-r" || a == "--resolution") { if (!parse_resolution(need("--resolution"), w, h)) { std::fprintf(stderr, "error: invalid resolution string\n"); return 2; } }
        else if (a == "-d" || a == "--max_depth") { const char *v = need("--max_depth"); char *e; long d = std::strtol(v, &e, 10); max_depth = (*e == 0) ? (int)d : 20; } // main.rs:21,87-97
        else if (a == "--dump-scene") dump_path = need("--dump-scene");
        else if (a == "--dump-scene-full") dump_full_path = need("--dump-scene-full");
        else if (a == "--default_lights") default_lights = true; // main.rs:48,99
        else if (a == "--env_map") env_map_path = need("--env_map"); // the Radiance .hdr --default_lights uses (the reference's bundled file is not shipped)
        else if (a == "--gpus") n_gpus = std::atoi(need("--gpus")); // not a flag of the reference: split the frame's rows over N devices of this process
        else if (a == "--even_bands") even_bands = true;               // equal-height bands instead of the 1-spp cost probe
        else if (a == "--preview") preview = true;                     // DIR/render.png is rewritten after every pass (the reference pushes the partial film to tev, headless.rs:197-214)
        else if (a == "--progress") progress = true;                   // a progress line per pass (the reference's progress bar, integrator.rs:631-634); like --preview it
                                                                       // renders through ptrs_render_progressive, which publishes the film after every pass (slower than the one-shot render)
        else if (a == "--headless") {}
        else if (a == "-c" || a == "--camera" || a == "-l" || a == "--log_level" || a == "-m" || a == "--module_log" || a == "--server") (void)need(a.c_str());
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "error: unknown flag %s\n", a.c_str()); return 2; }
        else scene_path = a;
    }
    if (scene_path.empty() || (!have_out && dump_path.empty() && dump_full_path.empty())) {
        std::fprintf(stderr, "usage: ptrs_headless SCENE(.xml|.gltf|.glb) -o DIR [-s SPP] [-r WxH] [-d DEPTH] [--default_lights --env_map FILE.hdr] [--gpus N [--even_bands]] [--preview] [--progress] [--headless]\n");
        return 2;
    }
    Camera camera; RenderScene scene; std::string err;
    if (!import_scene(scene_path, w, h, camera, scene, err, default_lights, env_map_path)) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    if (!dump_full_path.empty()) { if (!dump_scene_full(dump_full_path, camera, scene)) { std::fprintf(stderr, "error: cannot write %s\n", dump_full_path.c_str()); return 1; } if (!have_out && dump_path.empty()) return 0; }
    if (!dump_path.empty()) { if (!dump_scene(dump_path, camera, scene)) { std::fprintf(stderr, "error: cannot write %s\n", dump_path.c_str()); return 1; } if (!have_out) return 0; }
    int32_t sb[4];
    camera.film.get_sample_bounds(sb);
    PathIntegrator integrator(SamplerBuilder(spp, sb), max_depth, progress || preview);
    integrator.preprocess(scene);
    if (preview && have_out) integrator.on_pass = [&](uint32_t done, uint32_t total, int32_t, int32_t) {
        std::string e;
        if (done < total && !write_png_rgba8(out_dir + "/render.png", camera.film.width, camera.film.height, camera.film.to_rgba_image(), e)) std::fprintf(stderr, "\nwarning: preview: %s\n", e.c_str());
    };
    PtrsStats st{};
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (n_gpus > 1) {
        std::vector<PtrsStats> sts((size_t)n_gpus);
        std::vector<int32_t> bands;
        rc = integrator.render_multi(camera, scene, n_gpus, !even_bands, sts.data(), &bands);
        for (int d = 0; rc == PTRS_OK && d < n_gpus; ++d) {
            std::fprintf(stderr, "INFO device %d: rows %d..%d, %llu rays\n", d, bands[(size_t)d], bands[(size_t)d + 1], (unsigned long long)(sts[(size_t)d].rays_extension + sts[(size_t)d].rays_shadow + sts[(size_t)d].rays_mis));
            st.samples += sts[(size_t)d].samples; st.rays_extension += sts[(size_t)d].rays_extension; st.rays_shadow += sts[(size_t)d].rays_shadow; st.rays_mis += sts[(size_t)d].rays_mis;
        }
    } else rc = integrator.render(camera, scene, &st);
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != PTRS_OK) { std::fprintf(stderr, "error: render failed (%d): %s\n", rc, integrator.last_error.c_str()); return 1; }
    std::fprintf(stderr, "INFO rendering took: %.3fs (%llu samples, %llu rays, %.1f Mray/s)\n", secs, (unsigned long long)st.samples,
                 (unsigned long long)(st.rays_extension + st.rays_shadow + st.rays_mis), (double)(st.rays_extension + st.rays_shadow + st.rays_mis) / secs / 1e6);
    const std::string out = out_dir + "/render.png"; // main.rs:70
    if (!write_png_rgba8(out, camera.film.width, camera.film.height, camera.film.to_rgba_image(), err)) { std::fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
    std::fprintf(stderr, "INFO wrote %s\n", out.c_str());
    return 0;
}
