#!/usr/bin/env python3
"""Phase breakdown of k_shade from a diagnostic build with in-kernel stamps (tools/ablate.sh stamps "-DPTRS_STAMPS").

    PTRS_LIB=pathtracer-rs_amd/libptrs_stamps.so python tools/stamps.py [workload ...]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("pathtracer-rs_amd")
scenes = importlib.import_module("pathtracer-rs_amd.scenes")

NAMES = ["0 queue + path state loads", "1 triangle record + Sobol tables", "2 light record", "3 tri_surface + make_bsdf", "4 light_sample_li", "5 bsdf f/pdf + shadow ray",
         "6 BSDF sample + light pdf + MIS ray", "7 NEE stores + continuation sample + RR", "8 state stores", "9 (early returns) + queue pushes", "10 items (count)", "11 kernel tail"]
CFG = {"cornell": ((1024, 1024), 32), "colonnade": ((1280, 720), 16), "classroom": ((1920, 1080), 16)}
for w in (sys.argv[1:] or ["cornell"]):
    res, spp = CFG[w]
    cam, scene = pkg.import_scene(os.path.join(ROOT, "data", "cornell-box.xml"), res) if w == "cornell" else getattr(scenes, w)(res)
    integ = pkg.PathIntegrator(pkg.SamplerBuilder(spp, cam.film.get_sample_bounds()), 15)
    with pkg.options(lanes=1):
        integ.render(cam, scene, flags=pkg.abi.FLAG_TIMING)
    st = integ.last_stats
    d = list(st.debug)
    tot = sum(d[:10]) + d[11]
    print("%s: shade kernels %.1f ms, %d item-waves" % (w, st.ms_shade_kernels, d[10]))
    for k, nm in enumerate(NAMES):
        if k == 10:
            continue
        print("  %-48s %6.1f %%   %8.0f clocks per item-wave" % (nm, 100.0 * d[k] / max(tot, 1), d[k] / max(d[10], 1)))
