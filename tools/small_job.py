#!/usr/bin/env python3
"""A small render job several times (for API / kernel traces of what a render call costs around its kernels):
python tools/small_job.py [ROWS] [N] [PASSES: the job in that many sample chunks instead of the library's plan]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pkg = importlib.import_module("pathtracer-rs_amd")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cam, scene = pkg.import_scene(os.path.join(root, "data", "cornell-box.xml"), (1024, 1024))
integ = pkg.PathIntegrator(pkg.SamplerBuilder(256, cam.film.get_sample_bounds()), 15)
film = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda")
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if passes:
    st = integ.render_device(cam, scene, film.data_ptr(), stream=0, row_begin=448, row_end=448 + rows)
    per_spp = st.samples // 256
    integ.paths_per_pass = per_spp * ((256 + passes - 1) // passes)
for k in range(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    st = integ.render_device(cam, scene, film.data_ptr(), stream=0, row_begin=448, row_end=448 + rows)
    torch.cuda.synchronize(); print("call %d: %.2f ms (library: total %.2f, enqueue %.2f; %d launches, %d lanes, %d passes)" % (k, (time.perf_counter() - t) * 1e3, st.ms_total, st.ms_enqueue, st.kernel_launches, st.lanes, st.passes))
