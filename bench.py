#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render() hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: one process per GPU over RCCL.  Started under a launcher (python -m torch.distributed.run --nproc-per-node N ...
    bench.py --gpus N ..., RANK / WORLD_SIZE in the environment) it is one of the N ranks; started plainly it launches the N
    ranks ITSELF (self_launch: a child `python -m torch.distributed.run` before this process has imported torch or touched a
    GPU), relays rank 0's JSON line and exits with the child's code.

Workload (BASELINE.json configs[1]): data/cornell-box.xml, 1024x1024, 256 spp, max depth 15, Sobol sampler,
Lambertian + area light.  One *step* = one complete PathIntegrator::render of that frame with the film accumulators
resident in HBM.  With N GPUs the film's rows are split into N bands (strong scaling, each rank traces its band plus the
2-row filter halo); the bands are gathered on rank 0 with one RCCL collective inside the timed region.
`value` = BVH queries (extension + shadow + MIS rays) of the frame / step time; halo rows a band re-traces are not counted.

The timed steps run the library's default path (no per-launch events).  The JSON line carries
  roofline     -- per kernel class (traversal = k_extend* + k_connect*, shade = k_shade*): launch durations from HIP events
                  of ONE extra untimed frame on a single pipeline lane (a kernel has the GPU to itself), combined with the
                  hardware counters of the same kernels committed under profiles/ (rocprofv3 --pmc passes of
                  `bench.py --profile`, summarised by tools/prof_report.py; used only when the summary carries the hash
                  of the kernel sources this run was built from).  Every fraction is bounded by 1: HBM bytes against
                  8 TB/s; the vector-ALU time the instructions need against the MEASURED issue rates of gfx950
                  (profiles/r03_valu_ceiling.json: fp32 add / mul / fma issue beside one comparison / select / min-max
                  class instruction per ~4.2 cycles and SIMD), as a lo..hi pair because the counters split instructions
                  by kind, not by issue class; what a wave's cycles went to (issuing, s_waitcnt, waiting for a slot);
                  LDS-array cycles against CU cycles.  `bound` names what the numbers say for the class that takes the
                  most GPU time.  SURVEY 8(d)'s algorithmic bytes-per-ray figure is kept as `algorithmic` (it counts
                  BVH bytes that LDS / L2 serve, not HBM).
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
SIMDS, CUS = 1024.0, 256.0
CLASS_B_CYCLES = 4.2                       # profiles/r03_valu_ceiling.json: one comparison / select / min-max class instruction per 4.2 cycles and SIMD; fp32 add / mul / fma issue beside them
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_%s.json")
CEILING_FILE = os.path.join(ROOT, "profiles", "r03_valu_ceiling.json")
CALIB_FILE = os.path.join(ROOT, "profiles", "r04_fetch_calib.json")
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
    if name.startswith("k_tail"):
        return "tail"  # the fused late rounds: traversal and shading of thin segments in one kernel
    if name.startswith("k_film"):
        return "film"
    return "aux"


def load_pmc(workload, source_hash):
    """Per-class sums of the committed counter summary, or (None, meta) when it was measured on other kernel sources."""
    path = PMC_FILE % workload
    if not os.path.exists(path):
        return None, {"file": None}
    j = json.load(open(path))
    meta = dict(j.get("_meta", {}))
    meta["file"] = os.path.relpath(path, ROOT)
    meta["source_hash_now"] = source_hash
    meta["usable"] = meta.get("source_hash") == source_hash
    if not meta["usable"]:
        return None, meta
    frames = max(int(meta.get("frames", 1)), 1)
    cls = {}
    keys = ("hbm_bytes", "valu_insts", "valu_fp32_add_mul_fma", "valu_int", "salu_insts", "lds_insts")
    for name, k in j.items():
        if name.startswith("_"):
            continue
        c = cls.setdefault(kernel_class(name), dict(launches=0.0, lane_insts=0.0, total_ms=0.0, cycles=0.0, wave_cycles=0.0, issue=0.0, wait=0.0, stall=0.0, lds_cycles=0.0, lds_conflict=0.0,
                                                     nb_lo=0.0, nb_hi=0.0, kernels=[], **{q: 0.0 for q in keys}))
        c["launches"] += k["calls"] / frames
        for q in keys:
            c[q] += k.get(q, 0.0) / frames
        c["lane_insts"] += k["valu_insts"] * k["lanes_per_valu_inst"] / frames
        c["total_ms"] += k["total_ms"] / frames
        cyc = k["clock_ghz_profiled"] * 1e9 * k["total_ms"] * 1e-3  # cycles the kernel's dispatches were in flight in the profiled run
        c["cycles"] += cyc / frames
        wc = k["waves_resident_per_simd"] * cyc * SIMDS / 4.0       # SQ_WAVE_CYCLES (quad-cycles)
        c["wave_cycles"] += wc / frames
        c["issue"] += k["wave_issue_frac"] * wc / frames; c["wait"] += k["wave_wait_frac"] * wc / frames; c["stall"] += k["wave_stall_frac"] * wc / frames
        c["nb_lo"] += k["valu_class_b_share_lo"] * k["valu_insts"] / frames; c["nb_hi"] += k["valu_class_b_share_hi"] * k["valu_insts"] / frames
        if k.get("lds_busy_frac") is not None:
            c["lds_cycles"] += k["lds_busy_frac"] * cyc * CUS / frames
            c["lds_conflict"] += (k.get("lds_conflict_share") or 0.0) * k["lds_busy_frac"] * cyc * CUS / frames
        c["kernels"].append(name)
    return cls, meta


def class_roofline(name, ms_excl, launches_excl, pmc, algorithmic_bytes=None):
    """One class: durations live (single-lane frame), counters from the committed PMC summary (per frame, same workload, same sources)."""
    r = {"ms_per_frame_single_lane": ms_excl, "launches_per_frame": launches_excl, "avg_launch_ms_single_lane": ms_excl / max(launches_excl, 1)}
    if algorithmic_bytes is not None:
        r["algorithmic_gbs"] = algorithmic_bytes / (ms_excl * 1e-3) / 1e9 if ms_excl > 0 else 0.0
        r["algorithmic_frac_of_hbm_peak"] = r["algorithmic_gbs"] / HBM_PEAK_GBS
    if pmc is None or name not in pmc:
        r["counters"] = None
        return r
    c = pmc[name]
    ok = abs(c["launches"] - launches_excl) < 0.5  # the counters belong to this pipeline only if the launch counts agree too
    r["counters"] = {"file_launches_per_frame": c["launches"], "matches_live_launch_count": ok, "kernels": sorted(c["kernels"])}
    if ok and ms_excl > 0:
        sec = ms_excl * 1e-3
        clock = c["cycles"] / (c["total_ms"] * 1e-3)            # Hz, profiled run
        simd_cycles = sec * clock * SIMDS                        # SIMD-cycles of the live single-lane frame at the profiled clock
        nv = c["valu_insts"]
        r["hbm_bytes_per_launch"] = c["hbm_bytes"] / max(c["launches"], 1)
        r["hbm_counter_gbs"] = c["hbm_bytes"] / sec / 1e9
        r["hbm_counter_frac"] = r["hbm_counter_gbs"] / HBM_PEAK_GBS
        r["valu_insts_per_launch"] = nv / max(c["launches"], 1)
        r["valu_ginst_per_s"] = nv / sec / 1e9
        r["lanes_per_valu_inst"] = c["lane_insts"] / max(nv, 1.0)
        r["valu_class_b_share_lo_hi"] = [c["nb_lo"] / max(nv, 1.0), c["nb_hi"] / max(nv, 1.0)]
        r["valu_pipe_frac_lo_hi"] = [CLASS_B_CYCLES * max(c["nb_lo"], nv / 2.0) / simd_cycles, CLASS_B_CYCLES * max(c["nb_hi"], nv / 2.0) / simd_cycles]
        r["valu_pipe_frac"] = 0.5 * sum(r["valu_pipe_frac_lo_hi"])
        r["valu_ceiling_ginst_per_s"] = r["valu_ginst_per_s"] / max(r["valu_pipe_frac"], 1e-9)  # what the SIMDs could issue of this instruction mix
        r["wave_issue_frac"] = c["issue"] / max(c["wave_cycles"], 1.0); r["wave_wait_frac"] = c["wait"] / max(c["wave_cycles"], 1.0); r["wave_stall_frac"] = c["stall"] / max(c["wave_cycles"], 1.0)
        r["waves_resident_per_simd_profiled"] = 4.0 * c["wave_cycles"] / max(c["cycles"] * SIMDS, 1.0)
        r["salu_per_valu"] = c["salu_insts"] / max(nv, 1.0)
        r["lds_busy_frac"] = c["lds_cycles"] / max(c["cycles"] * CUS, 1.0) if c["lds_cycles"] else None
        r["lds_conflict_share"] = c["lds_conflict"] / c["lds_cycles"] if c["lds_cycles"] else None
        if r["hbm_counter_frac"] >= 0.5:
            r["bound"] = "hbm"
        elif r["valu_pipe_frac"] >= 0.6:
            r["bound"] = "valu-issue"
        elif r["lds_busy_frac"] and r["lds_busy_frac"] >= 0.6:
            r["bound"] = "lds"
        elif r["wave_wait_frac"] >= 0.5:
            r["bound"] = "waitcnt"
        else:
            r["bound"] = "mixed"
    return r


def collective_smoke():
    """A child process brings up the RCCL process group at world size 1 on this GPU and runs the film gather of the N-GPU path
    (parallel.gather_film_rows), so that the first multi-GPU run does not meet that code for the first time."""
    import subprocess
    code = ("import os, sys, importlib, time; sys.path.insert(0, %r); import torch, torch.distributed as dist\n"
            "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')\n"
            "torch.cuda.set_device(0); dev = torch.device('cuda', 0)\n"
            "t = time.time(); dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)\n"
            "p = torch.ones(1, device=dev); dist.all_reduce(p); torch.cuda.synchronize()\n"
            "par = importlib.import_module('pathtracer-rs_amd.parallel')\n"
            "film = torch.arange(64 * 32 * 4, dtype=torch.float32, device=dev).reshape(64, 32, 4); ref = film.clone()\n"
            "par.gather_film_rows(film, 64, 0, 1, bounds=[0, 64], mode='gather', force=True); torch.cuda.synchronize()\n"
            "assert torch.equal(film, ref); dist.barrier(); dist.destroy_process_group()\n"
            "print('ok %%.1f s' %% (time.time() - t))\n") % ROOT
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        last = (p.stdout.strip().splitlines() or [""])[-1]
        return {"ran": True, "ok": p.returncode == 0 and last.startswith("ok"), "detail": last if p.returncode == 0 else (p.stderr.strip()[-300:] or last),
                "what": "nccl (= RCCL) process group at world size 1 in a child process: init, all_reduce, parallel.gather_film_rows on a 64-row film, barrier, destroy"}
    except subprocess.TimeoutExpired:
        return {"ran": True, "ok": False, "detail": "timed out after 120 s"}
    except Exception as e:  # never let the smoke take the bench line down
        return {"ran": False, "ok": False, "detail": repr(e)}


def self_launch(argv, n):
    """`bench.py --gpus N` (N > 1) started without a launcher: this process -- which has not imported torch and will never touch a
    GPU -- starts `python -m torch.distributed.run` with N ranks of this script as a CHILD process (never an exec), relays what
    the ranks print (rank 0's JSON line) and returns the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env["PTRS_BENCH_LAUNCHED_BY"] = "bench.py self_launch"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    print("bench.py: --gpus %d without WORLD_SIZE: launching %d ranks: %s" % (n, n, " ".join(cmd)), file=sys.stderr)
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in p.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return p.wait()


def init_dist(torch, dist, dev, want):
    """Process group of a multi-rank run.  "nccl" is RCCL on ROCm; if it cannot be brought up (or fails its first all-reduce) the run
    stays measurable over gloo with host staging, and says so.  Returns the backend that actually runs."""
    import datetime
    tmo = datetime.timedelta(seconds=int(os.environ.get("PTRS_DIST_TIMEOUT_S", "300")))  # a collective that hangs ends the run instead of the driver's budget
    if want == "nccl":
        try:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)  # fail here, not in the timed region, if RCCL cannot be brought up
            torch.cuda.synchronize()
            return "nccl"
        except Exception as e:
            print("warning: RCCL initialisation failed (%r); falling back to gloo with host staging" % (e,), file=sys.stderr)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:
                pass
            want = "gloo"
    dist.init_process_group(want, timeout=tmo)
    return want


def rehearse_main(args, rank, world, launched_by):
    """--rehearse: the multi-rank plumbing of this script WITHOUT a render (the library has no CPU path, so a GPU-less box can
    run this and nothing else): launcher -> ranks -> process group -> band plan -> the default film gather -> one JSON line
    from rank 0.  Every rank fills its band of a small film with a pattern; rank 0 checks the gathered film.  No metric."""
    import torch
    import torch.distributed as dist
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    backend = os.environ.get("PTRS_DIST_BACKEND", "gloo")
    use_cuda = backend == "nccl" and torch.cuda.is_available()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(dev)
    if world > 1:
        backend = init_dist(torch, dist, dev, backend) if use_cuda else (dist.init_process_group(backend) or backend)
    H, W = 96, 8
    cost = [1.0 + (3.0 if 20 <= y < 40 else 0.0) for y in range(H)]  # uneven rows: the planned bands differ in height
    bounds = par.plan_bands(H, world, cost) if world > 1 else [0, H]
    want = torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    film[bounds[rank]:bounds[rank + 1]] = want[bounds[rank]:bounds[rank + 1]].to(dev)
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 1)):
        par.gather_film_rows(film, H, rank, world, bounds=bounds)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(film.cpu(), want)) if rank == 0 else True
    if rank == 0:
        print(json.dumps({"metric": "rehearsal of the multi-rank path (no render, no rays)", "value": None, "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / max(args.steps, 1) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "pattern", "rehearsal": True,
                          "config": {"workload": "none: band plan + film gather only", "row_bands": world, "band_plan": bounds,
                                     "dist": {"backend": "rccl" if backend == "nccl" else backend, "world_size": dist.get_world_size() if world > 1 else 1, "launched_by": launched_by}},
                          "gathered_film_ok": ok}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


TRACE_WORKLOADS = {"trace-colonnade": ("colonnade", (1280, 720)), "trace-classroom": ("classroom", (1920, 1080))}


def trace_main(args):
    """--workload trace-colonnade | trace-classroom: traversal alone on an HBM-resident tree (SURVEY 8(d)'s formula and north_star's
    "HBM-read roofline during traversal" mean something here: the tree does not fit LDS).  Ray sets of a real frame: the camera rays
    and the rays leaving the paths' third vertices (ptrs_render_dump_rays), traced with the frame's own extension kernel
    (ptrs_trace_bench); the reference's precedent is benches/benchmark_pathtracer.rs:35-54."""
    import numpy as np
    pkg = importlib.import_module("pathtracer-rs_amd")
    scenes = importlib.import_module("pathtracer-rs_amd.scenes")
    name, res = TRACE_WORKLOADS[args.workload]
    if args.node_order >= 0:
        pkg.set_option("node_order", args.node_order)  # read at scene creation
    cam, scene = getattr(scenes, name)(res)
    integ = pkg.PathIntegrator(pkg.SamplerBuilder(16, cam.film.get_sample_bounds()), DEPTH)
    integ.preprocess(scene)
    max_rays = 8 << 20
    if args.rays and os.path.exists(args.rays):  # (profiling runs: the ray sets come from a file, so that the profiled process launches the traced kernel only)
        z = np.load(args.rays)
        sets = {"camera rays (round 0)": z["camera"], "secondary rays (round 3)": z["secondary"]}
    else:
        sets = {"camera rays (round 0)": pkg.dump_rays(integ, cam, scene, 0, max_rays), "secondary rays (round 3)": pkg.dump_rays(integ, cam, scene, 3, max_rays)}
        if args.rays:
            np.savez(args.rays, camera=sets["camera rays (round 0)"], secondary=sets["secondary rays (round 3)"])
            print("saved %d + %d rays to %s" % (len(sets["camera rays (round 0)"]), len(sets["secondary rays (round 3)"]), args.rays))
            return
    if args.profile:
        st, _ = pkg.trace_bench(scene, sets["secondary rays (round 3)"], repeats=args.steps)
        print("profile: %d launches of %d secondary rays" % (args.steps + 1, len(sets["secondary rays (round 3)"])))
        return
    src_hash = pkg.build_id()  # the id the LOADED library was built with (sources + every compiler flag)
    pmc_path = PMC_FILE % (args.workload + ("-order%d" % args.node_order if args.node_order > 0 else ""))
    pmc = None
    if os.path.exists(pmc_path):
        j = json.load(open(pmc_path))
        if j.get("_meta", {}).get("source_hash") == src_hash:
            pmc = next((v for k, v in j.items() if k.startswith("k_extend_rf")), None)
    res_sets = {}
    for label, rays in sets.items():
        pkg.trace_bench(scene, rays, repeats=max(args.warmup, 1))
        st, hits = pkg.trace_bench(scene, rays, repeats=args.steps, want_hits=True)
        n = len(rays)
        check, _ = pkg.trace_rays(scene, rays[:200000])
        same = bool(np.array_equal(check["prim"], hits["prim"][:200000]) and np.array_equal(check["b1"].view(np.uint32), hits["b1"][:200000].view(np.uint32)))
        npr, tpr = st.nodes_visited / n, st.tris_tested / n
        b_ray = 32.0 + 32.0 * npr + 48.0 * tpr + 16.0
        sec = st.ms_trace * 1e-3 / args.steps
        res_sets[label] = {"rays": n, "ms_per_launch": sec * 1e3, "mray_per_s": n / sec / 1e6, "boxes_per_ray": npr, "tris_per_ray": tpr, "algorithmic_bytes_per_ray": b_ray,
                           "algorithmic_gbs": b_ray * n / sec / 1e9, "algorithmic_frac_of_hbm_peak": b_ray * n / sec / 1e9 / HBM_PEAK_GBS, "hits_equal_ptrs_trace_rays": same,
                           "hit_fraction": float((hits["prim"] >= 0).mean())}
    sec_set = res_sets["secondary rays (round 3)"]
    # `frac` is what HBM delivered (counters) against 8 TB/s -- the figure north_star's ">= 40 % of HBM-read roofline during traversal" asks
    # for; the algorithmic bytes of SURVEY 8(d) are kept beside it under their own name: they exceed what HBM can deliver because L2 and LDS
    # serve most of them.
    roof = {"kernel": "k_extend_rf (quad nodes, top of the tree in LDS, phase voting, lane refill, persistent waves)", "bound": None,
            "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "algorithmic": {"what": "SURVEY 8(d): 32 B ray in + 32 B per box tested + 48 B per triangle tested + 16 B hit out, x rays / kernel time; NOT HBM traffic (most of it is served by L2 and the LDS-cached top of the tree)",
                            "bytes_per_ray": sec_set["algorithmic_bytes_per_ray"], "gbs": sec_set["algorithmic_gbs"], "frac_of_hbm_peak": sec_set["algorithmic_frac_of_hbm_peak"]},
            "sets": res_sets}
    if pmc is not None:
        launches = pmc["calls"]
        roof["traffic"] = pmc["hbm_bytes"] / max(launches, 1)
        roof["achieved"] = pmc["hbm_gbs"]; roof["frac"] = pmc["hbm_counter_frac_of_8TBs"]; roof["l2_hit_rate"] = pmc["l2_hit_rate"]
        roof["bound"] = pmc.get("bound")
        roof["what"] = ("2 x FETCH_SIZE + WRITE_SIZE of the profiled launches (secondary-ray set) / their duration against the 8 TB/s HBM peak.  These are the bytes that crossed the fabric below L2 -- "
                        "HBM AND Infinity Cache: the counter cannot tell them apart (profiles/r04_fetch_calib.json) and this tree (tens of MB) fits the 256 MiB Infinity Cache, so the HBM share is smaller still.  "
                        "north_star's '>= 40 % of the HBM-read roofline during traversal' is therefore not evaluable as HBM on a BASELINE-sized scene; as memory-side bytes the kernel is just under it, and it is bound "
                        "by neither bandwidth: a dependent walk over random 128-byte records at this occupancy (calibration kernel) moves `random_record_walk_ceiling_gbs`, this kernel -- which computes four slab tests per record -- `achieved`")
        roof["fetch_size_note"] = "FETCH_SIZE x 2: calibrated on this access shape (tools/fetch_calib.hip): FETCH_SIZE = TCC_MISS x 64 B exactly, for 16 B/lane streams and for gathers of 128-byte records, Infinity-Cache-resident or not"
        roof["l2_miss_bytes_per_launch"] = (pmc.get("l2_miss_bytes") or 0.0) / max(launches, 1)
        if os.path.exists(CALIB_FILE):
            cj = json.load(open(CALIB_FILE))
            roof["random_record_walk_ceiling_gbs"] = {"infinity_cache_resident_table_64MB": cj["gather_64MB_dependent"]["record_visits_per_s"] * 128.0 / 1e9, "hbm_resident_table_2GB": cj["gather_2GB_dependent"]["record_visits_per_s"] * 128.0 / 1e9,
                                                      "streaming_read_2GB": cj["stream_2GB"]["gbs"], "what": "tools/fetch_calib.hip at 5 waves per SIMD: dependent visits of random 128-byte records x 128 B; a 16 B/lane stream"}
            roof["frac_of_random_record_walk_ceiling"] = (roof["achieved"] or 0.0) / roof["random_record_walk_ceiling_gbs"]["infinity_cache_resident_table_64MB"]
        roof["counters"] = {k: pmc.get(k) for k in ("valu_pipe_frac_lo", "valu_pipe_frac_hi", "wave_issue_frac", "wave_wait_frac", "wave_stall_frac", "waves_resident_per_simd", "lanes_per_valu_inst", "salu_per_valu", "calls")}
        roof["counters_from"] = os.path.relpath(pmc_path, ROOT)
    out = {"metric": "Mray/s, traversal only, %s (%d triangles), secondary rays of a %dx%d frame at depth 3" % (name, scene.num_triangles(), res[0], res[1]),
           "value": sec_set["mray_per_s"], "unit": "Mray/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec_set["ms_per_launch"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (procedural %s scene, seeded; rays dumped from its own render)" % name,
           "config": {"workload": "%s: k_extend_rf over the extension rays of round 0 and round 3 of a %dx%d, 16 spp frame (stand-in scene for BASELINE configs[%d])" % (args.workload, res[0], res[1], 2 if name == "colonnade" else 3),
                      "node_order": pkg.get_option("node_order"), "options": {k: pkg.get_option(k) for k in ("grid_mult", "refill", "vote", "stack_lds")}},
           "roofline": roof}
    print(json.dumps(out))


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
    ap.add_argument("--no-collective-smoke", action="store_true", help="N = 1: skip the child process that brings up RCCL at world size 1 and runs the film gather")
    ap.add_argument("--even-bands", action="store_true", help="N > 1: bands of equal height instead of equal cost")
    ap.add_argument("--rehearse", action="store_true", help="the multi-rank plumbing only (launch, process group, band plan, film gather), no render: runs without a GPU over gloo")
    ap.add_argument("--profile", action="store_true", help="for rocprofv3 runs: render exactly --steps frames on ONE pipeline lane (no counter / warm-up frames, no JSON)")
    ap.add_argument("--tail-at", type=int, default=-2, help="--profile: the round at which the profiled single-lane frames hand over to the fused tail (-1: no tail); tools/prof.sh passes what --print-tail-at printed, so that the profiled frame is launched like the bench's own single-lane frame")
    ap.add_argument("--profile-lanes", type=int, default=1, help="--profile: pipeline lanes of the profiled frames (1: a kernel has the GPU to itself; 0: the library's default)")
    ap.add_argument("--print-tail-at", action="store_true", help="prints the hand-over round of this workload's single-lane frame (after the frame that teaches the library the scene's survival profile) and exits")
    ap.add_argument("--rays", default="", help="trace workloads: an .npz of ray sets; written (and nothing else done) when it does not exist, read instead of rendering when it does")
    ap.add_argument("--node-order", type=int, default=-1, help="trace workloads: quad-node order behind the LDS-cached top (0 depth-first, 1 treelets)")
    ap.add_argument("--workload", default="cornell", choices=sorted(WORKLOADS) + sorted(TRACE_WORKLOADS),
                    help="cornell = BASELINE configs[1] (the headline); colonnade / classroom = synthetic stand-ins for configs[2] / [3]")
    args = ap.parse_args()
    if args.workload in TRACE_WORKLOADS:
        return trace_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:  # no launcher around us: be the launcher (before torch is imported, before any GPU call)
        sys.exit(self_launch(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched_by = os.environ.get("PTRS_BENCH_LAUNCHED_BY", "external launcher (RANK / WORLD_SIZE in the environment)" if world > 1 else "single process")
    if args.rehearse:
        sys.exit(rehearse_main(args, rank, world, launched_by))

    import numpy as np
    import torch
    import torch.distributed as dist

    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = os.environ.get("PTRS_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; "gloo" only for rehearsals that share a GPU
    if world > 1:
        backend = init_dist(torch, dist, dev, backend)

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
    # N > 1: bands of equal cost (per-row ray counts of ONE untimed 1-spp probe render with device counters; every rank computes the same
    # plan), unless --even-bands or the plan would take less than 1 % off the slowest band
    bounds, probe_ms, plan_gain = None, 0.0, None
    if world > 1 and not args.even_bands:
        cost = par.probe_row_cost(pkg, cam, scene, args.depth, device=local_rank, spp=1)  # (the process's first call into the library: scene upload and workspace allocation ride on it)
        torch.cuda.synchronize(); tp = time.perf_counter()
        par.probe_row_cost(pkg, cam, scene, args.depth, device=local_rank, spp=1, cache=False)  # what the probe itself costs (a host that plans a NEW view pays this once; the same view again comes out of the cache)
        torch.cuda.synchronize(); probe_ms = (time.perf_counter() - tp) * 1e3
        plan_gain = par.plan_gain(H, world, cost)
        if plan_gain >= 1.01:
            bounds = par.plan_bands(H, world, cost)
    if bounds is not None:
        row_b, row_e = bounds[rank], bounds[rank + 1]
    else:
        row_b, row_e = par.band_for_rank(H, rank, world)
    film = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    gather_mode, preflight = "p2p", None
    if world > 1:
        # The exact gather of the timed region (default mode, this run's band plan) once on a small film BEFORE anything is timed: the
        # first multi-GPU run meets the point-to-point batch here; if it raises, the padded dist.gather takes over and the line says so.
        small = torch.full((H, 4, 4), float(rank), dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
        try:
            par.gather_film_rows(small, H, rank, world, bounds=bounds)
            if backend == "nccl":
                torch.cuda.synchronize()
            if rank == 0:
                bb = bounds if bounds is not None else [par.band_for_rank(H, r, world)[0] for r in range(world)] + [H]
                assert all(bool((small[bb[r]:bb[r + 1]] == float(r)).all()) for r in range(world)), "gathered rows differ"
            preflight = "p2p ok"
        except Exception as e:  # (a hang instead ends at PTRS_DIST_TIMEOUT_S)
            preflight = "p2p failed: %r; using the padded gather" % (e,)
            gather_mode = "gather"
        flag = torch.tensor([1.0 if gather_mode == "gather" else 0.0], device=small.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)  # all ranks take the same branch
        if float(flag[0]) > 0:
            gather_mode = "gather"

    def step(flags):
        film.zero_()
        st = integ.render_device(cam, scene, film.data_ptr(), stream=stream, row_begin=row_b, row_end=row_e, flags=flags)
        if world > 1:
            if backend == "nccl":
                par.gather_film_rows(film, H, rank, world, bounds=bounds, mode=gather_mode)
            else:  # CPU-staged gather (rehearsal only)
                host = film.cpu()
                par.gather_film_rows(host, H, rank, world, bounds=bounds, mode=gather_mode)
                if rank == 0:
                    film.copy_(host)
        return st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.print_tail_at:
        step(0)  # (the scene's survival profile, as the bench's own first frame leaves it)
        with pkg.options(lanes=1):
            st = step(0)
            sync()
        print(int(st.tail_round) if st.tail_launches else -1)
        return
    if args.profile:
        pin = {} if args.tail_at == -2 else ({"tail": 0} if args.tail_at < 0 else {"tail_at": args.tail_at})
        with pkg.options(lanes=args.profile_lanes, **pin):
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
    # untimed: one frame on a single pipeline lane with an event pair around every launch, so that no two kernels share the
    # machine -- each kernel class's own duration
    with pkg.options(lanes=1):
        xst = step(pkg.abi.FLAG_TIMING)
        sync()
    for _ in range(args.warmup):
        step(0)
    sync()
    t0 = time.perf_counter()
    tot = dict(rays=0, samples=0)
    for _ in range(args.steps):
        st = step(0)  # the library's default path: no per-launch events
        tot["rays"] += st.rays
        tot["samples"] += st.samples
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
        src_hash = pkg.build_id()  # the id the LOADED library was built with (sources + every compiler flag)
        pmc, pmc_meta = load_pmc(args.workload, src_hash) if (world == 1 and standard) else (None, {"file": None, "why": "counters are per frame of the standard single-GPU workload"})
        classes = {
            "traversal": class_roofline("traversal", xst.ms_extend + xst.ms_connect, xst.extend_launches + xst.connect_launches, pmc, algorithmic_bytes=xst.rays * b_ray),
            "shade": class_roofline("shade", xst.ms_shade_kernels, xst.shade_launches, pmc),
            "tail": class_roofline("tail", xst.ms_tail, xst.tail_launches, pmc),
        }
        dom = max(classes, key=lambda k: classes[k]["ms_per_frame_single_lane"])
        d = classes[dom]
        ceiling = None
        if os.path.exists(CEILING_FILE):
            cj = json.load(open(CEILING_FILE))
            pick = lambda inst: max((r["per_cycle_per_simd"] for r in cj["rows"] if r["inst"] == inst and r["chains"].startswith("8")), default=None)
            ceiling = {"file": os.path.relpath(CEILING_FILE, ROOT), "per_cycle_per_simd": {k: pick(k) for k in ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32_e64 (mask in an SGPR pair)", "v_cmp_lt_f32_e64 -> SGPR pair", "v_fma_f64")},
                       "what": "measured wave64 issue rates on MI355X (tools/valu_ceiling.hip): fp32 add / mul / fma ~0.45 per cycle and SIMD, comparison / select / min-max class ~0.24; a mix of both classes issues side by side"}
        roof = {
            "kernel": "%s kernels (%s)" % (dom, ", ".join(d["counters"]["kernels"]) if d.get("counters") else ("k_extend_rf + k_connect_rf" if dom == "traversal" else "k_shade<material, features>")),
            "bound": d.get("bound"),
            "why": "neither class moves more than a fraction of the HBM peak (hbm.frac); the scarce resource is vector issue: `frac` is the share of the SIMDs' vector-ALU time the class's instructions need at the measured gfx950 issue rates (valu_ceiling), the rest of a wave's time is s_waitcnt (wave_wait_frac) -- dependent chains inside a wave (LDS reads -> slab test -> stack) that more resident waves did NOT shorten (DESIGN 4.1)",
            "achieved": d.get("valu_ginst_per_s"), "peak": d.get("valu_ceiling_ginst_per_s"), "unit": "G wave64 VALU instructions/s (peak = what 1024 SIMDs issue of this class's instruction mix at the measured rates)",
            "frac": d.get("valu_pipe_frac"), "frac_lo_hi": d.get("valu_pipe_frac_lo_hi"),
            "wave": {"issue": d.get("wave_issue_frac"), "wait": d.get("wave_wait_frac"), "stall": d.get("wave_stall_frac"), "salu_per_valu": d.get("salu_per_valu"), "lanes_per_valu_inst": d.get("lanes_per_valu_inst"),
                     "lds_busy": d.get("lds_busy_frac"), "lds_conflict_share": d.get("lds_conflict_share")},
            "traffic": d.get("hbm_bytes_per_launch"),
            "hbm": {"achieved": d.get("hbm_counter_gbs"), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d.get("hbm_counter_frac"), "what": "2 x FETCH_SIZE + WRITE_SIZE of the class's kernels (committed rocprofv3 PMC passes; FETCH_SIZE calibrated: profiles/r04_fetch_calib.json) / their single-lane duration measured in this run: bytes below L2 (HBM + Infinity Cache)"},
            "algorithmic": {"what": "SURVEY 8(d): 32 B ray in + 32 B per box tested + 48 B per triangle tested + 16 B hit out, x rays / traversal-kernel time; NOT HBM traffic (LDS- and L2-served)",
                            "bytes_per_ray": b_ray, "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray, "gbs": classes["traversal"].get("algorithmic_gbs"), "frac_of_hbm_peak": classes["traversal"].get("algorithmic_frac_of_hbm_peak"),
                            "bytes_per_launch": xst.rays * b_ray / max(xst.extend_launches + xst.connect_launches, 1)},
            "classes": classes, "valu_ceiling": ceiling,
            "counters_from": pmc_meta, "counters_usable": bool(d.get("valu_pipe_frac") is not None),
            "timing": "HIP events around every launch of ONE untimed frame on a single pipeline lane (ms_per_frame_single_lane); the timed steps run the default path on %d lanes without events" % int(st.lanes),
            "single_lane_frame_ms": {"extend": xst.ms_extend, "connect": xst.ms_connect, "shade": xst.ms_shade_kernels, "aux (generate, epilogue, resolve)": xst.ms_aux, "film": xst.ms_film, "tail (fused late rounds)": xst.ms_tail},
            "b_state_bytes_per_path_round": 216, "hbm_copy_measured_gbs": hbm_copy_gbs(torch, dev),
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
                       "band_plan": ("single band" if world == 1 else ("equal height" if bounds is None else "equal cost (one 1-spp probe render with a device counter per sample row): rows %s" % (bounds,))),
                       "band_plan_gain_predicted": plan_gain,
                       "band_probe_ms": probe_ms, "value_incl_band_probe": rays / (dt + args.steps * probe_ms * 1e-3) / 1e6,  # a host that plans its bands per frame pays the probe per frame
                       "dist": {"backend": ("none" if world == 1 else ("rccl" if backend == "nccl" else backend + " (fallback or rehearsal: NOT an RCCL number)")), "world_size": dist.get_world_size() if world > 1 else 1,
                                "launched_by": launched_by, "gather_mode": gather_mode if world > 1 else None, "gather_preflight": preflight,
                                "note": "no N > 1 hardware run existed when this was written: the first one is the first execution of the RCCL gather at world size > 1"},
                       "collective": ("none" if world == 1 else ("rccl gather of film row bands" if backend == "nccl" else backend + " gather (host-staged fallback, NOT an RCCL number)")),
                       "halo_overhead": rows_traced / float(H + 4) - 1.0, "rays_traced_incl_halo_per_step": rays_traced / args.steps,
                       "rays_by_kind_rank0_per_step": {"extension (closest hit; one per path vertex reached)": int(st.rays_extension), "shadow (any hit)": int(st.rays_shadow), "mis (closest hit)": int(st.rays_mis)},
                       "host_enqueue_ms_per_step": st.ms_enqueue,
                       "pipeline_lanes": int(st.lanes), "launch_share_of_resident_capacity_pct": int(st.grid_pct), "timed_path": "library default (no PTRS_FLAG_TIMING)",
                       "launch": {"queue_segments_per_pass": st.queue_segments, "passes_per_frame": st.passes, "kernel_launches_per_frame": st.kernel_launches,
                                  "workgroups_last_launch": dict(zip(("extend", "connect", "shade", "aux"), [int(x) for x in st.grid_wgs])),
                                  "resident_workgroups_per_cu": dict(zip(("extend", "connect", "shade", "aux"), [int(x) for x in st.resident_wgs_per_cu])),
                                  "fused_tail": {"launches_per_frame": int(st.tail_launches), "from_round": (int(st.tail_round) if st.tail_launches else None),
                                                 "single_lane_frame": {"launches": int(xst.tail_launches), "from_round": (int(xst.tail_round) if xst.tail_launches else None)}},
                                  "options": {k: pkg.get_option(k) for k in ("lanes", "grid_mult", "grid_pct", "persist", "refill", "refill_connect", "vote", "tail", "tail_at", "tail_paths")}}},
            "roofline": roof,
            "film_check": film_check, "film_check_rel_l2": roof_rel,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "cornell" and standard:
            gpu_rows = film[500:628].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(pkg, gpu_rows)
        if world == 1 and not args.no_collective_smoke:
            out["collective_smoke"] = collective_smoke()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
