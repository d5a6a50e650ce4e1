"""HOST TWIN binding -- TEST INFRASTRUCTURE ONLY (see twin.cpp)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
abi = importlib.import_module("pathtracer-rs_amd.abi")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", _HERE])
        L = C.CDLL(os.path.join(_HERE, "libtwin.so"))
        L.twin_last_error.restype = C.c_char_p
        if L.twin_load_tables(os.path.join(_ROOT, "data", "sobol_tables.bin").encode()) != 0:
            raise RuntimeError("twin: cannot load sobol tables")
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("twin error %d: %s" % (rc, lib().twin_last_error().decode()))


def round_up_pow2(v):
    return 1 << max(0, (int(v) - 1).bit_length())


class TwinScene:
    def __init__(self, render_scene, bvh=None):
        self._h = C.c_void_p()
        desc = render_scene.desc(bvh)
        _check(lib().twin_scene_create(C.byref(desc), C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                lib().twin_scene_destroy(self._h)
        except Exception:
            pass

    def render(self, camera, params, want_samples=False, film=None):
        W, H = params.width, params.height
        if film is None:
            film = np.zeros((H, W), dtype=abi.FILM_DTYPE)
        spp = params.spp if params.sampler == abi.SAMPLER_STRATIFIED else round_up_pow2(params.spp)
        samples = np.zeros((H + 4, W + 4, spp, 3), dtype=np.float32) if want_samples else None
        stats = abi.PtrsStats()
        cam = camera.to_abi()
        _check(lib().twin_render(self._h, C.byref(cam), C.byref(params), C.c_void_p(film.ctypes.data),
                                 C.c_void_p(samples.ctypes.data) if want_samples else None, C.byref(stats)))
        return film, samples, stats

    def trace_rays(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        hits = np.zeros(rays.shape[0], dtype=abi.HIT_DTYPE)
        stats = abi.PtrsStats()
        _check(lib().twin_trace_rays(self._h, rays.shape[0], C.c_void_p(rays.ctypes.data), int(any_hit), C.c_void_p(hits.ctypes.data), C.byref(stats)))
        return hits, stats


def sobol_samples(params, px, py, sample_nums, dims):
    px = np.ascontiguousarray(px, dtype=np.int32)
    py = np.ascontiguousarray(py, dtype=np.int32)
    sn = np.ascontiguousarray(sample_nums, dtype=np.uint64)
    dm = np.ascontiguousarray(dims, dtype=np.uint32)
    out = np.zeros(px.shape[0], dtype=np.float32)
    idx = np.zeros(px.shape[0], dtype=np.uint64)
    _check(lib().twin_sobol_samples(C.byref(params), px.shape[0], C.c_void_p(px.ctypes.data), C.c_void_p(py.ctypes.data), C.c_void_p(sn.ctypes.data),
                                    C.c_void_p(dm.ctypes.data), C.c_void_p(out.ctypes.data), C.c_void_p(idx.ctypes.data)))
    return out, idx
