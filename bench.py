#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): data/cornell-box.xml, 1024x1024, 256 spp, max depth 15, Sobol sampler,
Lambertian + area light.  One *step* = one complete PathIntegrator::render of that frame with the film accumulators
resident in HBM.  With N GPUs the film's rows are split into N bands (strong scaling, each rank traces its band plus the
2-row filter halo); the bands are gathered on rank 0 with one RCCL collective inside the timed region.
`value` = BVH queries (extension + shadow + MIS rays) of the frame / step time; halo rows a band re-traces are not counted.

The JSON line carries
  roofline     -- per kernel class (traversal = k_extend* + k_connect*, shade = k_shade*): launch durations from HIP events
                  (inside the timed steps, where passes overlap on several pipeline lanes, AND from one extra frame on a
                  single lane, where a kernel has the GPU to itself -- the fractions use the latter), combined with the
                  per-launch hardware counters of the same kernels committed under profiles/ (rocprofv3 --pmc passes of
                  `bench.py --profile`, summarised by tools/prof_report.py): HBM bytes (FETCH_SIZE + WRITE_SIZE), VALU
                  instructions and active lanes.  Top-level fields describe the class that takes the most GPU time.
                  These kernels are bound by VALU issue under divergence, not by HBM: `bound` says so, the HBM fraction
                  from the counters is reported beside it, and SURVEY 8(d)'s algorithmic bytes-per-ray figure is kept
                  as `algorithmic` (it counts BVH bytes that LDS / L2 serve, not HBM).
  cpu_baseline -- the oracle (C++ restatement of the reference's CPU path, kind "port") timed on this box's host
                  cores on a bounded band of the same frame.
  film_check   -- rows of the film the timed steps produced, compared with a committed oracle fixture
                  (tests/golden/bench_*_rows.npz, made by tests/golden/make_golden.py).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEPTH = 15
SCENE = os.path.join(ROOT, "data", "cornell-box.xml")
WORKLOADS = {
    "cornell": dict(res=(1024, 1024), spp=256, metric="Cornell Box",
                    label="cornell-box %dx%d spp=%d max_depth=%d, Lambertian + area light (BASELINE configs[1])"),
    "colonnade": dict(res=(1280, 720), spp=64, metric="colonnade (Sponza-class stand-in, %d triangles)",
                      label="colonnade %dx%d spp=%d max_depth=%d, Disney metal + image texture + punctual lights (stand-in for BASELINE configs[2]; the Sponza glTF is not available offline)"),
    "classroom": dict(res=(1920, 1080), spp=128, metric="classroom (Classroom-class stand-in, %d triangles)",
                      label="classroom %dx%d spp=%d max_depth=%d, glass + Disney dielectric + HDR environment light data/abandoned_tank_farm_04_1k.hdr (stand-in for BASELINE configs[3]; the Classroom glTF is not available offline)"),
}
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E peak
VALU_PEAK_TLANEOPS = 256 * 4 * 64 * 2.4e9 / 2.0 / 1e12  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz = 78.6 T lane-ops/s (157 TFLOP/s as FMA)
PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_%s.json")
FIXTURE = os.path.join(ROOT, "tests", "golden", "bench_%s_rows.npz")


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


def cpu_baseline(pkg, gpu_rows, rows=128, row0=500):
    """Oracle (CPU port of the reference path) on rows [row0, row0+rows) of the Cornell frame; also checks the GPU's
    film rows against the ones it produces (the oracle as the checker, never as the thing shipped)."""
    import numpy as np
    from oracle import orc

    W, H = WORKLOADS["cornell"]["res"]
    spp = WORKLOADS["cornell"]["spp"]
    cam, scene = pkg.import_scene(SCENE, (W, H))
    o = orc.OracleScene(scene)
    cores = host_cores()
    p = orc.make_params(W, H, spp, DEPTH, row_begin=row0, row_end=row0 + rows)
    t = time.time()
    film, _, st = o.render(cam, p, n_threads=cores)
    dt = time.time() - t
    rays = st.rays_extension + st.rays_shadow + st.rays_mis
    # the same port on one thread (the reference's `disable_rayon` feature), 2 rows
    p1 = orc.make_params(W, H, spp, DEPTH, row_begin=row0, row_end=row0 + 2)
    t = time.time()
    _, _, st1 = o.render(cam, p1, n_threads=1)
    dt1 = time.time() - t
    rays1 = st1.rays_extension + st1.rays_shadow + st1.rays_mis
    out = {
        "value": rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port", "os_cpu_count": os.cpu_count(),
        "single_thread_mray_s": rays1 / dt1 / 1e6, "single_thread_sample": "rows %d..%d, %d rays, %.1f s" % (row0, row0 + 2, rays1, dt1),
        "msample_per_s": st.samples / dt / 1e6,
        "sample": "output rows %d..%d of the 1024x1024/256spp/depth-15 Cornell frame (%d li() samples, %d rays, %.1f s, %d threads, 16x16 tiles, dynamic queue)"
                  % (row0, row0 + rows, st.samples, rays, dt, cores),
        "nodes_per_ray": st.nodes_visited / max(rays, 1), "tris_per_ray": st.tris_tested / max(rays, 1),
    }
    if gpu_rows is not None:
        ref = np.concatenate([film["rgb"], film["weight"][..., None]], axis=-1)[row0:row0 + rows].astype(np.float64)
        got = gpu_rows.astype(np.float64)
        out["film_rel_l2_vs_gpu_rows"] = float(np.sqrt(((got - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-30)))
    return out


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


def kernel_class(name):
    if name.startswith("k_extend") or name.startswith("k_connect") or name.startswith("k_trace"):
        return "traversal"
    if name.startswith("k_shade"):
        return "shade"
    if name.startswith("k_film"):
        return "film"
    return "aux"


def load_pmc(workload):
    """Per-class sums of the committed counter summary: {class: {launches, hbm_bytes, valu_insts, thread_cycles, ...}}, meta."""
    path = PMC_FILE % workload
    if not os.path.exists(path):
        return None, None
    j = json.load(open(path))
    meta = j.get("_meta", {})
    frames = max(int(meta.get("frames", 1)), 1)
    cls = {}
    for name, k in j.items():
        if name.startswith("_"):
            continue
        c = cls.setdefault(kernel_class(name), dict(launches=0.0, hbm_bytes=0.0, valu_insts=0.0, lane_insts=0.0, busy_ms=0.0, total_ms=0.0, kernels=[]))
        c["launches"] += k["calls"] / frames
        c["hbm_bytes"] += k["hbm_bytes"] / frames
        c["valu_insts"] += k["valu_insts"] / frames
        c["lane_insts"] += k["valu_insts"] * k["lanes_per_valu_inst"] / frames
        c["busy_ms"] += k.get("valu_busy_frac", 0.0) * k["total_ms"] / frames
        c["total_ms"] += k["total_ms"] / frames
        c["kernels"].append(name)
    return cls, meta


def class_roofline(name, ms_excl, launches_excl, ms_timed, launches_timed, pmc, algorithmic_bytes=None):
    """One class: durations live, counters from the committed PMC summary (per frame, same workload)."""
    r = {"ms_per_frame_single_lane": ms_excl, "launches_per_frame": launches_excl,
         "avg_launch_ms_single_lane": ms_excl / max(launches_excl, 1), "avg_launch_ms_timed_overlapped": ms_timed / max(launches_timed, 1)}
    if algorithmic_bytes is not None:
        r["algorithmic_gbs"] = algorithmic_bytes / (ms_excl * 1e-3) / 1e9 if ms_excl > 0 else 0.0
        r["algorithmic_frac"] = r["algorithmic_gbs"] / HBM_PEAK_GBS
    if pmc is None or name not in pmc:
        r["counters"] = None
        return r
    c = pmc[name]
    ok = abs(c["launches"] - launches_excl) < 0.5  # the counters belong to this pipeline only if the launch counts agree
    r["counters"] = {"file_launches_per_frame": c["launches"], "matches_live_launch_count": ok, "kernels": sorted(c["kernels"])}
    if ok and ms_excl > 0:
        sec = ms_excl * 1e-3
        r["hbm_bytes_per_launch"] = c["hbm_bytes"] / max(c["launches"], 1)
        r["hbm_counter_gbs"] = c["hbm_bytes"] / sec / 1e9
        r["hbm_counter_frac"] = r["hbm_counter_gbs"] / HBM_PEAK_GBS
        r["valu_insts_per_launch"] = c["valu_insts"] / max(c["launches"], 1)
        r["lanes_per_valu_inst"] = c["lane_insts"] / max(c["valu_insts"], 1.0)
        r["valu_issue_frac"] = (c["valu_insts"] / sec) / (VALU_PEAK_TLANEOPS * 1e12 / 64.0)
        r["valu_tlaneops"] = c["lane_insts"] / sec / 1e12
        r["valu_lane_frac"] = r["valu_tlaneops"] / VALU_PEAK_TLANEOPS
        # share of the profiled kernel time the SIMDs' VALU pipes were executing (4 x SQ_ACTIVE_INST_VALU / SIMDs / GRBM cycles, from the
        # profiled run): a plain fp32 wave64 instruction occupies the pipe for 4 cycles, only packed / dual-issued ones reach the 2-cycle peak
        r["valu_pipe_busy_frac_profiled"] = c["busy_ms"] / c["total_ms"] if c["total_ms"] > 0 else None
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--res", type=int, default=0, help="cornell only: square resolution")
    ap.add_argument("--depth", type=int, default=DEPTH)
    ap.add_argument("--paths-per-pass", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--even-bands", action="store_true", help="N > 1: bands of equal height instead of equal cost")
    ap.add_argument("--profile", action="store_true", help="for rocprofv3 runs: render exactly --steps frames on ONE pipeline lane (no counter / warm-up frames, no JSON)")
    ap.add_argument("--workload", default="cornell", choices=sorted(WORKLOADS),
                    help="cornell = BASELINE configs[1] (the headline); colonnade / classroom = synthetic stand-ins for configs[2] / [3]")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
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
    wl = WORKLOADS[args.workload]
    W, H = wl["res"]
    spp = args.spp or wl["spp"]
    if args.workload == "cornell":
        if args.res:
            W = H = args.res
        cam, scene = pkg.import_scene(SCENE, (W, H))
    else:
        scenes = importlib.import_module("pathtracer-rs_amd.scenes")
        cam, scene = getattr(scenes, args.workload)((W, H))
    standard = (spp, (W, H), args.depth, args.paths_per_pass) == (wl["spp"], wl["res"], DEPTH, 0)
    integ = pkg.PathIntegrator(pkg.SamplerBuilder(spp, cam.film.get_sample_bounds()), args.depth, device=local_rank, paths_per_pass=args.paths_per_pass)
    integ.preprocess(scene)
    # N > 1: bands of equal cost (per-row ray counts of an untimed 1-spp probe; every rank computes the same plan), unless --even-bands
    bounds, probe_ms = None, 0.0
    if world > 1 and not args.even_bands:
        tp = time.perf_counter()
        bounds = par.plan_bands(H, world, par.probe_row_cost(pkg, cam, scene, args.depth, device=local_rank, strips=64, spp=1))
        probe_ms = (time.perf_counter() - tp) * 1e3
        row_b, row_e = bounds[rank], bounds[rank + 1]
    else:
        row_b, row_e = par.band_for_rank(H, rank, world)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step(flags):
        film.zero_()
        st = integ.render_device(cam, scene, film.data_ptr(), stream=stream, row_begin=row_b, row_end=row_e, flags=flags)
        if world > 1:
            if backend == "nccl":
                par.gather_film_rows(film, H, rank, world, bounds=bounds)
            else:  # CPU-staged gather (rehearsal only)
                host = film.cpu()
                par.gather_film_rows(host, H, rank, world, bounds=bounds)
                if rank == 0:
                    film.copy_(host)
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.profile:
        with pkg.options(lanes=1):
            for _ in range(args.steps):
                st = step(0)
            sync()
        print("profile frames: %d, rays per frame %d, kernel launches per frame %d" % (args.steps, st.rays, st.kernel_launches))
        return
    lanes = pkg.get_option("lanes")
    # untimed: counters for the algorithmic per-ray node / triangle averages
    cst = step(pkg.abi.FLAG_COUNTERS)
    c_rays = cst.rays
    nodes_per_ray, tris_per_ray = cst.nodes_visited / max(c_rays, 1), cst.tris_tested / max(c_rays, 1)
    # untimed: one frame on a single pipeline lane, so that no two kernels share the machine -- each kernel class's own
    # duration (the timed steps below overlap passes on several lanes, which stretches every kernel's span)
    with pkg.options(lanes=1):
        xst = step(pkg.abi.FLAG_TIMING)
        sync()
    for _ in range(args.warmup):
        step(pkg.abi.FLAG_TIMING)
    sync()
    t0 = time.perf_counter()
    tot = dict(rays=0, samples=0, ms_extend=0.0, ms_connect=0.0, ms_shade_kernels=0.0, ms_aux=0.0, ms_film=0.0, extend_launches=0, connect_launches=0, shade_launches=0)
    for _ in range(args.steps):
        st = step(pkg.abi.FLAG_TIMING)
        tot["rays"] += st.rays
        tot["samples"] += st.samples
        for k in ("ms_extend", "ms_connect", "ms_shade_kernels", "ms_aux", "ms_film", "extend_launches", "connect_launches", "shade_launches"):
            tot[k] += getattr(st, k)
    sync()
    dt = time.perf_counter() - t0

    # halo: band k traces sample rows [rb, re + 4) but owns [rb, re) (+ the last 4 for the last band): count owned rows only
    traced_rows = (row_e - row_b) + 4
    owned_rows = (row_e - row_b) + (4 if rank == world - 1 else 0)
    own = owned_rows / traced_rows
    vals = torch.tensor([dt, tot["rays"] * own, tot["samples"] * own, float(tot["rays"]), float(traced_rows)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = vals.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        rays, samples, rays_traced, rows_traced = float(sm[1]), float(sm[2]), float(sm[3]), float(sm[4])
    else:
        rays, samples, rays_traced, rows_traced = float(vals[1]), float(vals[2]), float(vals[3]), float(vals[4])
    if rank == 0:
        b_ray = 32.0 + 32.0 * nodes_per_ray + 48.0 * tris_per_ray + 16.0
        pmc, pmc_meta = load_pmc(args.workload) if (world == 1 and standard) else (None, None)
        steps = args.steps
        classes = {
            "traversal": class_roofline("traversal", xst.ms_extend + xst.ms_connect, xst.extend_launches + xst.connect_launches,
                                        (tot["ms_extend"] + tot["ms_connect"]) / steps, (tot["extend_launches"] + tot["connect_launches"]) / steps, pmc,
                                        algorithmic_bytes=xst.rays * b_ray),
            "shade": class_roofline("shade", xst.ms_shade_kernels, xst.shade_launches, tot["ms_shade_kernels"] / steps, tot["shade_launches"] / steps, pmc),
        }
        dom = max(classes, key=lambda k: classes[k]["ms_per_frame_single_lane"])
        d = classes[dom]
        have = d.get("valu_lane_frac") is not None
        roof = {
            "kernel": "%s kernels (%s)" % (dom, ", ".join(d["counters"]["kernels"]) if d.get("counters") else ("k_extend_rf + k_connect[_rf]" if dom == "traversal" else "k_shade<material, features>")),
            "bound": "valu-issue",
            "why": "divergent, latency-exposed scalar code: neither class moves more than a fraction of the HBM peak (hbm_counter_frac) and the tree / path state it reads is served by LDS and L2; what is scarce is VALU issue slots with lanes in them",
            "achieved": d.get("valu_tlaneops"), "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s (active-lane VALU instructions; peak = 1 wave64 instruction / 2 cycles / SIMD at 2.4 GHz = the 157 TFLOP/s fp32 vector peak)",
            "frac": d.get("valu_lane_frac"),
            "valu_pipe_busy_frac_profiled": d.get("valu_pipe_busy_frac_profiled"),  # the VALU pipes are busy this share of the time; `frac` is lower by the idle lanes and the 4-cycle plain-fp32 issue
            "traffic": d.get("hbm_bytes_per_launch"),
            "hbm": {"achieved": d.get("hbm_counter_gbs"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d.get("hbm_counter_frac"), "what": "FETCH_SIZE + WRITE_SIZE of the class's kernels (committed rocprofv3 PMC passes) / their single-lane duration measured in this run"},
            "algorithmic": {"what": "SURVEY 8(d): 32 B ray in + 32 B per box tested + 48 B per triangle tested + 16 B hit out, x rays / traversal-kernel time; NOT HBM traffic (LDS- and L2-served)",
                            "bytes_per_ray": b_ray, "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray, "gbs": classes["traversal"].get("algorithmic_gbs"), "frac_of_hbm_peak": classes["traversal"].get("algorithmic_frac"),
                            "bytes_per_launch": xst.rays * b_ray / max(xst.extend_launches + xst.connect_launches, 1)},
            "classes": classes,
            "counters_from": (os.path.relpath(PMC_FILE % args.workload, ROOT) if pmc is not None else None), "counters_meta": pmc_meta,
            "counters_usable": bool(have),
            "timing": "HIP events around every launch on the lanes' own streams; fractions use the single-lane frame (ms_per_frame_single_lane), the timed steps overlap passes on %d pipeline lanes" % lanes,
            "single_lane_frame_ms": {"extend": xst.ms_extend, "connect": xst.ms_connect, "shade": xst.ms_shade_kernels, "aux (generate, epilogue, resolve)": xst.ms_aux, "film": xst.ms_film},
            "b_state_bytes_per_path_round": 224, "hbm_copy_measured_gbs": hbm_copy_gbs(torch, dev),
        }
        # film check: rows of the timed film against the committed oracle fixture
        film_check = "no fixture for these settings"
        gpu_rows = None
        fx = FIXTURE % args.workload
        if standard and os.path.exists(fx):
            z = np.load(fx)
            r0, r1 = int(z["row0"]), int(z["row1"])
            got = film[r0:r1].cpu().numpy().astype(np.float64)
            ref = z["film"].astype(np.float64)
            rel = float(np.sqrt(((got - ref) ** 2).sum() / max((ref ** 2).sum(), 1e-30)))
            film_check = "ok" if rel < 1e-5 else "MISMATCH"
            roof_rel = rel
        else:
            roof_rel = None
        out = {
            "metric": "Mray/s, %s %dx%d, %d spp, depth %d (Msample/s in config)" % (wl["metric"] % scene.num_triangles() if "%d" in wl["metric"] else wl["metric"], W, H, spp, args.depth),
            "value": rays / dt / 1e6, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (%s, deterministic Sobol sequence)" % ("data/cornell-box.xml as parsed" if args.workload == "cornell" else "procedural %s scene, seeded" % args.workload),
            "config": {"workload": wl["label"] % (W, H, spp, args.depth),
                       "msample_per_s": samples / dt / 1e6, "rays_per_sample": rays / max(samples, 1.0), "row_bands": world,
                       "band_plan": ("single band" if world == 1 else ("equal height" if bounds is None else "equal cost (1-spp probe in 64 strips, %.0f ms, untimed setup): rows %s" % (probe_ms, bounds))),
                       "collective": ("none" if world == 1 else ("rccl gather of film row bands" if backend == "nccl" else backend + " gather (host-staged fallback, NOT an RCCL number)")),
                       "halo_overhead": rows_traced / float(H + 4) - 1.0, "rays_traced_incl_halo_per_step": rays_traced / args.steps,
                       "pipeline_lanes": lanes},
            "roofline": roof,
            "film_check": film_check, "film_check_rel_l2": roof_rel,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "cornell" and standard:
            gpu_rows = film[500:628].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(pkg, gpu_rows)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
