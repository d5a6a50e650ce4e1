"""Host mirror of the reference's integrator API over the HIP library.

Mirrors (reference file:line):
  SamplerBuilder::new(log, spp, &sample_bounds)          src/pathtracer/sampler/sobol.rs:35-60
  PathIntegrator::{new, preprocess, render,
                   render_single_pixel}                  src/pathtracer/integrator.rs:230,250,536,505
Construction matches src/main.rs:103-110:
    integrator = PathIntegrator(SamplerBuilder(spp, camera.film.get_sample_bounds()), max_depth)
    integrator.preprocess(scene); integrator.render(camera, scene)
render() accumulates into camera.film like the reference (callers clear() first for a fresh image).
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class PtrsError(RuntimeError):
    pass


def load_library():
    """Loads libptrs_hip.so (built in-tree by build.py).  Fails loudly when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("PTRS_LIB") or os.path.join(_HERE, "libptrs_hip.so")  # PTRS_LIB: A/B builds of the same library
    if not os.path.exists(so):
        raise PtrsError("HIP library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                        "There is no CPU fallback for the render path." % so)
    L = C.CDLL(so)
    L.ptrs_last_error.restype = C.c_char_p
    L.ptrs_build_id.restype = C.c_char_p
    if L.ptrs_abi_version() != 4:
        raise PtrsError("ABI version mismatch")
    structs = [abi.PtrsTexture, abi.PtrsMaterial, abi.PtrsMesh, abi.PtrsLight, abi.PtrsBvhNode, abi.PtrsSceneDesc, abi.PtrsCamera,
               abi.PtrsRenderParams, abi.PtrsStats, abi.PtrsHit]
    for i, s in enumerate(structs):
        if L.ptrs_abi_sizeof(i) != C.sizeof(s):
            raise PtrsError("ABI struct %s: library %d bytes, binding %d bytes" % (s.__name__, L.ptrs_abi_sizeof(i), C.sizeof(s)))
    L.ptrs_set_option.argtypes = [C.c_char_p, C.c_int64]
    L.ptrs_scene_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ptrs_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    _LIB = L
    # A/B convenience of this Python host only (the library itself reads no environment): PTRS_OPT_<NAME>=<int>
    for k, v in os.environ.items():
        if k.startswith("PTRS_OPT_"):
            set_option(k[len("PTRS_OPT_"):].lower(), int(v))
    return L


def set_option(name, value):
    """ptrs_set_option: process-wide tuning knob (lanes, refill, refill_connect, vote, stack_lds, grid_mult, node_form,
    workspace_pct); none of them changes a result."""
    _check(load_library().ptrs_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int64()
    _check(load_library().ptrs_get_option(name.encode(), C.byref(v)))
    return v.value


class options:
    """Context manager: `with options(lanes=1, refill=0): ...` sets knobs and restores the previous values."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def _check(rc):
    if rc != 0:
        raise PtrsError("ptrs error %d: %s" % (rc, _LIB.ptrs_last_error().decode() if _LIB is not None else "library not loaded"))


def round_up_pow2(v):
    return 1 << max(0, (int(v) - 1).bit_length())


class SamplerBuilder:
    """SobolSamplerBuilder (sobol.rs:25-78): spp is rounded up to a power of two with a warning."""

    def __init__(self, samples_per_pixel, sample_bounds):
        self.samples_per_pixel = round_up_pow2(samples_per_pixel)
        if self.samples_per_pixel != samples_per_pixel:
            warnings.warn("non power-of-two sample count rounded up to %d for sobol sampler" % self.samples_per_pixel)
        self.sample_bounds = tuple(sample_bounds)
        ext = max(self.sample_bounds[2] - self.sample_bounds[0], self.sample_bounds[3] - self.sample_bounds[1])
        self.resolution = round_up_pow2(ext)
        self.log_2_resolution = self.resolution.bit_length() - 1

    def with_seed(self, _seed):  # sobol.rs:75-77: a no-op in the reference as well
        return self


class StratifiedSamplerBuilder:
    """StratifiedSamplerBuilder::new(log, dim_pixel_samples, n_sampled_dimensions) (sampler/stratified.rs:22-36): spp =
    dim_pixel_samples^2, jittered.  The reference compiles this sampler but never builds one (sampler/mod.rs:169-170); here
    it selects PtrsRenderParams.sampler = PTRS_SAMPLER_STRATIFIED.  with_seed is applied per tile by render (integrator.rs:553)."""

    def __init__(self, dim_pixel_samples, n_sampled_dimensions):
        self.dim_pixel_samples = int(dim_pixel_samples)
        self.n_sampled_dimensions = int(n_sampled_dimensions)
        self.samples_per_pixel = self.dim_pixel_samples * self.dim_pixel_samples

    def with_seed(self, _seed):
        return self


class _DeviceScene:
    def __init__(self, render_scene, device=0, bvh=None):
        self.handle = C.c_void_p()
        self.device = device
        desc = render_scene.desc(bvh)
        _check(load_library().ptrs_scene_create(C.byref(desc), int(device), C.byref(self.handle)))

    def info(self):
        n, d, t = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load_library().ptrs_scene_info(self.handle, C.byref(n), C.byref(d), C.byref(t)))
        return dict(bvh_nodes=n.value, bvh_max_depth=d.value, n_tris=t.value)

    def set_option(self, name, value):
        """ptrs_scene_set_option: this scene's renders take `value` for the knob instead of the process-wide setting."""
        _check(load_library().ptrs_scene_set_option(self.handle, name.encode(), C.c_int64(int(value))))

    def close(self):
        if self.handle:
            load_library().ptrs_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_scene(render_scene, device=0, bvh=None):
    key = "_ptrs_dev_%d" % device
    ds = getattr(render_scene, key, None)
    if ds is None or bvh is not None:
        ds = _DeviceScene(render_scene, device, bvh)
        setattr(render_scene, key, ds)
    return ds


class PathIntegrator:
    """PathIntegrator (integrator.rs:219-246): rr_threshold 1.0, rr_start_depth 3, rr_enable true."""

    def __init__(self, sampler_builder, max_depth, show_progress_bar=False, device=0, paths_per_pass=0):
        self.sampler_builder = sampler_builder
        self.max_depth = int(max_depth)
        self.rr_threshold, self.rr_start_depth, self.rr_enable = 1.0, 3, True
        self.show_progress_bar = show_progress_bar
        self.device = device
        self.paths_per_pass = paths_per_pass
        self.last_stats = None

    def preprocess(self, scene):  # integrator.rs:250-258
        if len(scene.lights) > 16:
            warnings.warn("scene contains too many lights for path integrator to handle well")

    def toggle_progress_bar(self):  # integrator.rs:260-262
        self.show_progress_bar = not self.show_progress_bar

    def params(self, camera, row_begin=0, row_end=0, flags=0):
        p = abi.PtrsRenderParams()
        p.width, p.height = camera.film.width, camera.film.height
        p.spp, p.max_depth = self.sampler_builder.samples_per_pixel, self.max_depth
        p.rr_threshold, p.rr_start_depth, p.rr_enable = self.rr_threshold, self.rr_start_depth, int(self.rr_enable)
        p.row_begin, p.row_end = row_begin, (row_end if row_end else camera.film.height)
        p.device, p.paths_per_pass, p.flags = self.device, self.paths_per_pass, flags
        if isinstance(self.sampler_builder, StratifiedSamplerBuilder):
            p.sampler, p.n_sampled_dimensions = abi.SAMPLER_STRATIFIED, self.sampler_builder.n_sampled_dimensions
        return p

    def render(self, camera, scene, row_begin=0, row_end=0, flags=0, want_samples=False):
        """integrator.rs:536-642 on the GPU; accumulates into camera.film.pixels (host memory)."""
        ds = _device_scene(scene, self.device)
        p = self.params(camera, row_begin, row_end, flags)
        cam = camera.to_abi()
        stats = abi.PtrsStats()
        film = camera.film.pixels
        samples = None
        if want_samples:
            samples = np.zeros((p.height + 4, p.width + 4, p.spp if p.sampler == abi.SAMPLER_STRATIFIED else round_up_pow2(p.spp), 3), dtype=np.float32)
        _check(load_library().ptrs_render_samples(ds.handle, C.byref(cam), C.byref(p), C.c_void_p(film.ctypes.data),
                                                  C.c_void_p(samples.ctypes.data) if want_samples else None, C.byref(stats)))
        self.last_stats = stats
        return samples if want_samples else None

    def render_device(self, camera, scene, film_device_ptr, stream=0, row_begin=0, row_end=0, flags=0):
        """Same, film accumulators in device memory (film_device_ptr: width*height*16 bytes)."""
        ds = _device_scene(scene, self.device)
        p = self.params(camera, row_begin, row_end, flags)
        cam = camera.to_abi()
        stats = abi.PtrsStats()
        _check(load_library().ptrs_render_device(ds.handle, C.byref(cam), C.byref(p), C.c_void_p(int(film_device_ptr)), C.c_void_p(int(stream)), C.byref(stats)))
        self.last_stats = stats
        return stats

    def render_progressive(self, camera, scene, on_pass, row_begin=0, row_end=0):
        """ptrs_render_progressive: like render(), and after every pass of the pipeline the rows it touched are copied into
        camera.film.pixels and on_pass(passes_done, passes_total, row_begin, row_end) is called (the preview hook the
        reference's headless front-end gets by polling its film, headless.rs:197-214)."""
        ds = _device_scene(scene, self.device)
        p = self.params(camera, row_begin, row_end)
        cam = camera.to_abi()
        stats = abi.PtrsStats()
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32)
        cb = cb_t(lambda _u, done, total, y0, y1: on_pass(done, total, y0, y1))
        _check(load_library().ptrs_render_progressive(ds.handle, C.byref(cam), C.byref(p), C.c_void_p(camera.film.pixels.ctypes.data), cb, None, C.byref(stats)))
        self.last_stats = stats
        return stats

    def set_scene_option(self, scene, name, value):
        """A tuning knob for this scene on this integrator's device only (ptrs_scene_set_option)."""
        _device_scene(scene, self.device).set_option(name, value)

    def render_multi(self, camera, scene, devices, bounds=None, row_cost=None, film_is_zero=False, scene_options=None):
        """ptrs_render_multi: the frame's rows split over `devices` (one PtrsScene per entry, one host thread each, bands
        gathered on the first device); bounds = n+1 row numbers, or planned by ptrs_plan_bands (weighted by row_cost when
        given).  Accumulates into camera.film.pixels; returns (bounds, [PtrsStats per device])."""
        L = load_library()
        n = len(devices)
        scenes = [_DeviceScene(scene, d) for d in devices]  # fresh replicas, also when several share a device
        try:
            for k, v in (scene_options or {}).items():
                for sc_ in scenes:
                    sc_.set_option(k, v)
            p = self.params(camera, flags=abi.FLAG_FILM_ZERO if film_is_zero else 0)
            cam = camera.to_abi()
            b = (C.c_int32 * (n + 1))()
            if bounds is None:
                rc = None if row_cost is None else np.ascontiguousarray(row_cost, dtype=np.float32)
                _check(L.ptrs_plan_bands(p.height, n, C.c_void_p(rc.ctypes.data) if rc is not None else None, b))
            else:
                b[:] = [int(v) for v in bounds]
            handles = (C.c_void_p * n)(*[s.handle for s in scenes])
            stats = (abi.PtrsStats * n)()
            _check(L.ptrs_render_multi(handles, n, C.byref(cam), C.byref(p), b, C.c_void_p(camera.film.pixels.ctypes.data), stats))
            return list(b), list(stats)
        finally:
            for s in scenes:
                s.close()

    def render_single_pixel(self, camera, pixel, scene):  # integrator.rs:505-534
        ds = _device_scene(scene, self.device)
        p = self.params(camera)
        cam = camera.to_abi()
        out = np.zeros((round_up_pow2(p.spp), 3), dtype=np.float32)
        self.last_single_pixel_paths = round_up_pow2(p.spp)
        _check(load_library().ptrs_render_single_pixel(ds.handle, C.byref(cam), C.byref(p), int(pixel[0]), int(pixel[1]), C.c_void_p(out.ctypes.data)))
        return out


def trace_rays(scene, rays, any_hit=False, device=0, bvh=None):
    """RenderScene::intersect / intersect_p (pathtracer/mod.rs:92-98) for a batch of rays
    (n x 7: o, d, t_max) through the traversal kernel."""
    ds = _device_scene(scene, device, bvh)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
    hits = np.zeros(rays.shape[0], dtype=abi.HIT_DTYPE)
    stats = abi.PtrsStats()
    _check(load_library().ptrs_trace_rays(ds.handle, rays.shape[0], C.c_void_p(rays.ctypes.data), int(any_hit), C.c_void_p(hits.ctypes.data), C.byref(stats)))
    return hits, stats


def dump_rays(integrator, camera, scene, round_no, max_rays):
    """ptrs_render_dump_rays: the extension rays of round `round_no` of the first pass of `integrator`'s render (n x 7: o, d, t_max)."""
    ds = _device_scene(scene, integrator.device)
    p = integrator.params(camera)
    cam = camera.to_abi()
    out = np.zeros((int(max_rays), 7), dtype=np.float32)
    n = C.c_uint32(0)
    _check(load_library().ptrs_render_dump_rays(ds.handle, C.byref(cam), C.byref(p), int(round_no), int(max_rays), C.c_void_p(out.ctypes.data), C.byref(n)))
    return out[: n.value].copy()


def trace_bench(scene, rays, repeats=5, device=0, want_hits=False):
    """ptrs_trace_bench: closest-hit traversal of `rays` with the frame's extension kernel, `repeats` timed launches.  Returns
    (stats, hits or None); stats.ms_trace is the sum of the timed launches, nodes_visited / tris_tested belong to one launch."""
    ds = _device_scene(scene, device)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
    hits = np.zeros(rays.shape[0], dtype=abi.HIT_DTYPE) if want_hits else None
    stats = abi.PtrsStats()
    _check(load_library().ptrs_trace_bench(ds.handle, rays.shape[0], C.c_void_p(rays.ctypes.data), int(repeats), C.c_void_p(hits.ctypes.data) if want_hits else None, C.byref(stats)))
    return stats, hits


def sobol_samples(params, px, py, sample_nums, dims):
    """SobolSampler::sample_dimension (sobol.rs:177-193) for arbitrary (pixel, sample, dimension)."""
    px = np.ascontiguousarray(px, dtype=np.int32)
    py = np.ascontiguousarray(py, dtype=np.int32)
    sn = np.ascontiguousarray(sample_nums, dtype=np.uint64)
    dm = np.ascontiguousarray(dims, dtype=np.uint32)
    out = np.zeros(px.shape[0], dtype=np.float32)
    idx = np.zeros(px.shape[0], dtype=np.uint64)
    _check(load_library().ptrs_sobol_samples(C.byref(params), px.shape[0], C.c_void_p(px.ctypes.data), C.c_void_p(py.ctypes.data), C.c_void_p(sn.ctypes.data),
                                             C.c_void_p(dm.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(idx.ctypes.data)))
    return out, idx


def selftest_div3(mode, n_sets, seed=1, device=0):
    """ptrs_selftest_div3: the device code's shared-divisor division against the compiler's IEEE division over ~n_sets generated operand
    sets.  Returns (mismatches, sets that took the fast path, first mismatch as 10 uint32 words: a0 a1 a2 b | got x3 | want x3)."""
    L = load_library()
    L.ptrs_selftest_div3.argtypes = [C.c_int32, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p]
    bad, fast = C.c_uint64(0), C.c_uint64(0)
    first = np.zeros(10, dtype=np.uint32)
    _check(L.ptrs_selftest_div3(int(device), int(mode), int(n_sets), int(seed), C.byref(bad), C.byref(fast), C.c_void_p(first.ctypes.data)))
    return int(bad.value), int(fast.value), first


def build_id():
    """ptrs_build_id: hash of the kernel sources and compiler flags the loaded library was built from (build.source_hash)."""
    return load_library().ptrs_build_id().decode()
