#!/usr/bin/env python3
"""Lane occupancy of the traversal kernels' two phases (PTRS_FLAG_COUNTERS), per workload and option set.

    python tools/occupancy.py [workload ...]      (options via PTRS_OPT_* as usual)
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("pathtracer-rs_amd")
scenes = importlib.import_module("pathtracer-rs_amd.scenes")
import numpy as np  # noqa: E402

CFG = {"cornell": ((1024, 1024), 16), "colonnade": ((1280, 720), 8), "classroom": ((1920, 1080), 8)}
for w in (sys.argv[1:] or list(CFG)):
    res, spp = CFG[w]
    cam, scene = pkg.import_scene(os.path.join(ROOT, "data", "cornell-box.xml"), res) if w == "cornell" else getattr(scenes, w)(res)
    integ = pkg.PathIntegrator(pkg.SamplerBuilder(spp, cam.film.get_sample_bounds()), 15)
    for opts in (dict(vote=0, refill_connect=16), dict(vote=1, refill_connect=16)):
        with pkg.options(lanes=1, **opts):
            cam.film.clear()
            integ.render(cam, scene, flags=pkg.abi.FLAG_COUNTERS)
            st = integ.last_stats
            ns, nv, ts, nt = st.node_steps_x64 / 64.0, st.node_visits, st.tri_steps_x64 / 64.0, st.tris_tested
            print("%-10s %-34s rays %9d  node steps/ray*64 %.1f occupancy %.3f | tri steps/ray*64 %.1f occupancy %.3f | visits/ray %.2f tris/ray %.2f" % (
                w, opts, st.rays, 64.0 * ns / st.rays, nv / max(64.0 * ns, 1), 64.0 * ts / st.rays, nt / max(64.0 * ts, 1), nv / st.rays, nt / st.rays))
