#!/usr/bin/env python3
"""Phase breakdown of the extension kernel from a diagnostic build with in-kernel stamps:

    tools/ablate.sh xstamps "-DPTRS_STAMPS_EXT" && PTRS_LIB=pathtracer-rs_amd/libptrs_xstamps.so python tools/stamps_ext.py [workload ...]

One single-lane frame with the fused tail off (every round through k_extend_rf); the wave clocks are summed over all waves of all launches."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("pathtracer-rs_amd")
scenes = importlib.import_module("pathtracer-rs_amd.scenes")

NAMES = ["0 retire (hit stores) + refill decision", "1 refill: ray loads until the rays are set up", "2 traversal steps (node visits / triangle tests)", "3 epilogue (hits -> emission, buckets)", "4 tickets + segment count", "5 -"]
FULL = bool(int(os.environ.get("FULL", "0")))  # FULL=1: the BASELINE sample counts (the per-segment costs then weigh what they weigh in a bench frame)
TAIL = int(os.environ.get("TAIL", "0"))
CFG = {"cornell": ((1024, 1024), 256 if FULL else 32), "colonnade": ((1280, 720), 64 if FULL else 16), "classroom": ((1920, 1080), 128 if FULL else 16)}
for w in (sys.argv[1:] or ["cornell"]):
    res, spp = CFG[w]
    cam, scene = pkg.import_scene(os.path.join(ROOT, "data", "cornell-box.xml"), res) if w == "cornell" else getattr(scenes, w)(res)
    integ = pkg.PathIntegrator(pkg.SamplerBuilder(spp, cam.film.get_sample_bounds()), 15)
    with pkg.options(lanes=1, tail=TAIL):
        if TAIL:
            integ.render(cam, scene)  # (the scene's survival profile: the measured frame then hands over to the fused tail where a bench frame does)
            cam.film.clear()
        integ.render(cam, scene, flags=pkg.abi.FLAG_TIMING)
    st = integ.last_stats
    d = list(st.debug)
    tot = sum(d[:5])
    print("%s: extension kernels %.1f ms; %d rays in %d refill batches (%.1f rays per batch), %d wave-steps, %.1f lanes with a ray per step (%.1f of them at a node), %d segments" % (
        w, st.ms_extend, d[8], d[7], d[8] / max(d[7], 1), d[6], d[9] / max(d[6], 1), d[11] / max(d[6], 1), d[10]))
    for k in range(5):
        print("  %-52s %6.1f %%   %9.0f clocks per refill batch   %7.1f per ray" % (NAMES[k], 100.0 * d[k] / max(tot, 1), d[k] / max(d[7], 1), d[k] / max(d[8], 1)))
    print("  wave-steps per ray %.2f; clocks per wave-step %.0f" % (d[6] / max(d[8], 1) * 1.0, d[2] / max(d[6], 1)))
