#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): Cornell Box 1024x1024, 256 spp, max depth 15, Sobol sampler,
Lambertian + area light.  One *step* = one complete PathIntegrator::render of that frame with the
film accumulators resident in HBM.  With N GPUs the film's rows are split into N bands (strong
scaling, each rank traces its band plus the 2-row filter halo); the bands are gathered on rank 0
with one RCCL collective inside the timed region.  `value` = all rays of the frame / step time.

The JSON line carries:
  roofline     -- dominant kernel = BVH traversal (k_trace): algorithmic bytes per ray
                  (32 ray in + 32 per node visited + 48 per triangle tested + 16 hit out, the
                  per-ray node/triangle counts measured by device counters in an untimed render of
                  the same frame) x rays / summed k_trace time from HIP events recorded on the
                  render stream inside the timed steps;  peak = 8 TB/s HBM3E.
  cpu_baseline -- the oracle (C++ restatement of the reference's CPU path, kind "port") timed on
                  this box's host cores on a bounded band of the same frame.
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SPP, DEPTH = 1024, 1024, 256, 15
SCENE = os.path.join(ROOT, "tests", "golden", "cornell-box.xml")
HBM_PEAK_GBS = 8000.0
# HBM bytes per traversal-kernel launch (k_extend + k_connect) for the DEFAULT workload on one GPU, from two
# separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; profiles/r01_v8_pmc_fetch_write.csv, tools/summarize_pmc.py; single lane):
# (83.36e6 + 255.52e6 KB fetched + 88.53e6 + 44.12e6 KB written) / 288 launches.  FETCH_SIZE is taken as
# reported (gfx950 under-reports wide streaming reads by 2x; these are 16-byte gathers, uncalibrated).
TRAFFIC_PMC_DEFAULT = (83.36e6 + 255.52e6 + 88.53e6 + 44.12e6) * 1024.0 / 288.0


def host_cores():
    """Cores this process may actually use: min(os.cpu_count, affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(round(q / per))))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(pkg, rows=128):
    """Oracle (CPU port of the reference path) on rows [500, 500+rows) of the same frame."""
    from oracle import orc

    cam, scene = pkg.import_scene(SCENE, (WIDTH, HEIGHT))
    o = orc.OracleScene(scene)
    cores = host_cores()
    p = orc.make_params(WIDTH, HEIGHT, SPP, DEPTH, row_begin=500, row_end=500 + rows)
    t = time.time()
    _, _, st = o.render(cam, p, n_threads=cores)
    dt = time.time() - t
    rays = st.rays_extension + st.rays_shadow + st.rays_mis
    # the same port on one thread (the reference's `disable_rayon` feature), 2 rows
    p1 = orc.make_params(WIDTH, HEIGHT, SPP, DEPTH, row_begin=500, row_end=502)
    t = time.time()
    _, _, st1 = o.render(cam, p1, n_threads=1)
    dt1 = time.time() - t
    rays1 = st1.rays_extension + st1.rays_shadow + st1.rays_mis
    return {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port", "os_cpu_count": os.cpu_count(),
        "single_thread_mray_s": rays1 / dt1 / 1e6, "single_thread_sample": "rows 500..502, %d rays, %.1f s" % (rays1, dt1),
        "msample_per_s": st.samples / dt / 1e6,
        "sample": "output rows 500..%d of the 1024x1024/256spp/depth-15 Cornell frame (%d li() samples, %d rays, %.1f s, %d threads, 16x16 tiles, dynamic queue)"
                  % (500 + rows, st.samples, rays, dt, cores),
        "nodes_per_ray": st.nodes_visited / max(rays, 1), "tris_per_ray": st.tris_tested / max(rays, 1),
    }


def hbm_copy_gbs(torch, dev, nbytes=1 << 32, reps=5):
    """Measured device-to-device copy rate (read + write bytes / time): the practical HBM ceiling on this box."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--res", type=int, default=WIDTH)
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--paths-per-pass", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="cornell", choices=["cornell", "colonnade", "classroom"],
                    help="cornell = BASELINE configs[1] (the headline); colonnade = synthetic stand-in for configs[2] (Sponza glTF is not available offline): 1280x720, 64 spp")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("PTRS_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for rehearsals that share a GPU
    if world > 1:
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)  # fail here, not in the timed region, if RCCL cannot be brought up
                torch.cuda.synchronize()
            except Exception as e:  # keep the run measurable: CPU-staged gather over gloo, flagged in the JSON
                print("warning: RCCL initialisation failed (%r); falling back to gloo with host staging" % (e,), file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"
                dist.init_process_group("gloo")
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("pathtracer-rs_amd")
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    if args.workload == "colonnade":
        scenes = importlib.import_module("pathtracer-rs_amd.scenes")
        W, H = 1280, 720
        if args.spp == SPP:
            args.spp = 64
        cam, scene = scenes.colonnade((W, H))
    elif args.workload == "classroom":
        scenes = importlib.import_module("pathtracer-rs_amd.scenes")
        W, H = 1920, 1080
        if args.spp == SPP:
            args.spp = 128
        cam, scene = scenes.classroom((W, H))
    else:
        W = H = args.res
        cam, scene = pkg.import_scene(SCENE, (W, H))
    integ = pkg.PathIntegrator(pkg.SamplerBuilder(args.spp, cam.film.get_sample_bounds()), args.depth, device=local_rank, paths_per_pass=args.paths_per_pass)
    integ.preprocess(scene)
    row_b, row_e = par.band_for_rank(H, rank, world)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(flags):
        film.zero_()
        st = integ.render_device(cam, scene, film.data_ptr(), stream=stream, row_begin=row_b, row_end=row_e, flags=flags)
        if world > 1:
            if backend == "nccl":
                par.gather_film_rows(film, H, rank, world)
            else:  # CPU-staged gather (rehearsal only)
                host = film.cpu()
                par.gather_film_rows(host, H, rank, world)
                if rank == 0:
                    film.copy_(host)
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: counters for the roofline's per-ray node / triangle averages, then warm-up
    cst = step(pkg.abi.FLAG_COUNTERS)
    c_rays = cst.rays
    nodes_per_ray, tris_per_ray = cst.nodes_visited / max(c_rays, 1), cst.tris_tested / max(c_rays, 1)
    # untimed: one frame on a single pipeline lane, so that no two kernels share the machine -- the traversal kernels'
    # own duration (the timed steps below overlap passes on several lanes, which stretches every kernel's span)
    lanes_env = os.environ.get("PTRS_LANES")
    os.environ["PTRS_LANES"] = "1"
    xst = step(pkg.abi.FLAG_TIMING)
    sync()
    if lanes_env is None:
        del os.environ["PTRS_LANES"]
    else:
        os.environ["PTRS_LANES"] = lanes_env
    x_ms_trace, x_launches, x_rays = xst.ms_trace, xst.trace_launches, xst.rays
    for _ in range(args.warmup):
        step(pkg.abi.FLAG_TIMING)
    sync()
    t0 = time.perf_counter()
    rays = samples = 0
    ms_trace = ms_shade = ms_film = 0.0
    trace_launches = 0
    for _ in range(args.steps):
        st = step(pkg.abi.FLAG_TIMING)
        rays += st.rays
        samples += st.samples
        ms_trace += st.ms_trace
        ms_shade += st.ms_shade
        ms_film += st.ms_film
        trace_launches += st.trace_launches
    sync()
    dt = time.perf_counter() - t0

    vals = torch.tensor([dt, float(rays), float(samples), ms_trace, ms_shade, ms_film, float(trace_launches), nodes_per_ray * c_rays, tris_per_ray * c_rays, float(c_rays)], dtype=torch.float64,
                        device=dev if backend == "nccl" else "cpu")
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        rays, samples = float(sm[1]), float(sm[2])
        ms_trace = float(mx[3])
        nodes_per_ray, tris_per_ray = float(sm[7]) / float(sm[9]), float(sm[8]) / float(sm[9])
        rays_rank0 = float(vals[1])
    else:
        rays_rank0 = float(rays)
    if rank == 0:
        b_ray = 32.0 + 32.0 * nodes_per_ray + 48.0 * tris_per_ray + 16.0
        # rank 0's own kernel: its rays x algorithmic bytes / its summed k_trace time
        achieved = (rays_rank0 * b_ray) / (float(vals[3]) * 1e-3) / 1e9 if float(vals[3]) > 0 else 0.0
        out = {
            "metric": "Mray/s, %s %dx%d, %d spp, depth %d (Msample/s in config)" % ("Cornell Box" if args.workload == "cornell" else "%s (%s-class stand-in, %d triangles)" % (args.workload, "Sponza" if args.workload == "colonnade" else "Classroom", scene.num_triangles()), W, H, args.spp, args.depth),
            "value": rays / dt / 1e6, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (data/cornell-box.xml as parsed, deterministic Sobol sequence)",
            "config": {"workload": ("cornell-box %dx%d spp=%d max_depth=%d, Lambertian + area light (BASELINE configs[1])" if args.workload == "cornell" else ("colonnade %dx%d spp=%d max_depth=%d, Disney metal + image texture + punctual lights (stand-in for BASELINE configs[2])" if args.workload == "colonnade" else "classroom %dx%d spp=%d max_depth=%d, glass + Disney dielectric + HDR environment light (stand-in for BASELINE configs[3])")) % (W, H, args.spp, args.depth),
                       "msample_per_s": samples / dt / 1e6, "rays_per_sample": rays / max(samples, 1.0), "row_bands": world, "collective": ("none" if world == 1 else ("rccl gather of film row bands" if backend == "nccl" else backend + " gather (host-staged fallback)")),
                       "ms_trace_per_step": ms_trace / args.steps, "ms_shade_per_step": float(vals[4]) / args.steps, "ms_film_per_step": float(vals[5]) / args.steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (TRAFFIC_PMC_DEFAULT if (world == 1 and args.workload == "cornell" and (args.spp, args.res, args.depth, args.paths_per_pass) == (SPP, WIDTH, DEPTH, 0)) else None),
                         "algorithmic_bytes_per_launch": rays_rank0 * b_ray / max(float(vals[6]), 1.0),
                         "kernel": "the BVH traversal kernels: k_extend_rf + k_connect_rf (k_extend + k_connect where lane refill is off); bytes and time summed over both", "bytes_per_ray": b_ray, "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray,
                         "avg_launch_ms": float(vals[3]) / max(float(vals[6]), 1.0), "launches": int(float(vals[6])),
                         # path-state traffic of the wavefront design, reported apart from the traversal figure
                         "timing": "HIP events around every traversal launch inside the timed steps; passes overlap on %s pipeline lanes there, so a launch's span includes time it shared the GPU with the other lanes' kernels" % (os.environ.get("PTRS_LANES") or "3"),
                         "exclusive": {"what": "the same kernels in one untimed frame on a single lane (nothing overlapped)", "achieved": (x_rays * b_ray) / (x_ms_trace * 1e-3) / 1e9 if x_ms_trace > 0 else 0.0,
                                       "frac": ((x_rays * b_ray) / (x_ms_trace * 1e-3) / 1e9 / HBM_PEAK_GBS) if x_ms_trace > 0 else 0.0, "avg_launch_ms": x_ms_trace / max(x_launches, 1)},
                         "b_state_bytes_per_path_round": 224, "hbm_copy_measured_gbs": hbm_copy_gbs(torch, dev)},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "cornell":
            out["cpu_baseline"] = cpu_baseline(pkg)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
