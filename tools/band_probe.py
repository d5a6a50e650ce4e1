#!/usr/bin/env python3
"""Cost of each of the N row bands of the Cornell frame on one GPU (how even is the multi-GPU split, and how much of
the single-GPU efficiency a band 1/N the size keeps).  python tools/band_probe.py [N] [spp]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pkg = importlib.import_module("pathtracer-rs_amd")
par = importlib.import_module("pathtracer-rs_amd.parallel")
cam, scene = pkg.import_scene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "cornell-box.xml"), (1024, 1024))
integ = pkg.PathIntegrator(pkg.SamplerBuilder(spp, cam.film.get_sample_bounds()), 15)
film = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda")
for _ in range(2):  # warm-up (workspace allocation, clocks)
    integ.render_device(cam, scene, film.data_ptr(), stream=0, row_begin=0, row_end=1024 // N)
torch.cuda.synchronize()
integ.render_device(cam, scene, film.data_ptr(), stream=0); torch.cuda.synchronize()  # (the first full frame grows the workspace from the warm-up band's: not timed -- round 3's and this round's first figures were, which overstated the ratio below)
fulls = []
for _ in range(3):
    t0 = time.perf_counter(); st = integ.render_device(cam, scene, film.data_ptr(), stream=0); torch.cuda.synchronize(); fulls.append(time.perf_counter() - t0)
full = min(fulls)
ts = []
for r in range(N):
    b, e = par.band_for_rank(1024, r, N)
    torch.cuda.synchronize(); t = time.perf_counter()
    s = integ.render_device(cam, scene, film.data_ptr(), stream=0, row_begin=b, row_end=e)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t, s.rays))
mx = max(t for t, _ in ts)
print("full frame %.1f ms; %d bands: %s ms; max/mean %.3f; speed-up if bands ran on %d GPUs: %.2f (ideal %d)" % (
    full * 1e3, N, [round(t * 1e3, 1) for t, _ in ts], mx / (sum(t for t, _ in ts) / N), N, full / mx, N))
