"""pathtracer-rs_amd -- MI355X-native wavefront path-tracing backend for pathtracer-rs's
PathIntegrator::render hot path.

The package is a thin host layer over the C ABI of include/ptrs.h (libptrs_hip.so: hand-written
HIP kernels for gfx950).  There is no CPU fallback: without the HIP library or without a GPU the
render entry points raise.
"""
from . import abi  # noqa: F401
from .integrator import PathIntegrator, build_id, PtrsError, SamplerBuilder, StratifiedSamplerBuilder, dump_rays, get_option, load_library, options, set_option, selftest_div3, sobol_samples, trace_bench, trace_rays  # noqa: F401
from .scene import Camera, Film, RenderScene, import_scene, look_at_camera  # noqa: F401
