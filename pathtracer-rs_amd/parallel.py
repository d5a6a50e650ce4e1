"""Multi-GPU sharding of the film: one process per GPU, contiguous bands of
output rows per rank, one gather of the bands to rank 0 over RCCL (torch.distributed backend
"nccl") at the end of the render.  Each sample depends only on (pixel, sample index) -- the sampler
ignores the tile seed (sampler/sobol.rs:75-77) -- so ranks exchange nothing while tracing; a rank
re-traces the 2-row filter halo on each side of its band (film.rs:60-106) so that its rows are
complete and bit-identical to the single-GPU render.  Bands are equal in height (band_for_rank) or equal in
cost (plan_bands over probe_row_cost: rows differ in work -- Cornell's most expensive eighth costs 1.11 x
the mean -- and the frame is done when the slowest rank is)."""
import torch
import torch.distributed as dist


def band_for_rank(height, rank, world):
    """Rows [begin, end) owned by `rank`: as even as possible, earlier ranks take the remainder."""
    base, rem = divmod(int(height), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def plan_bands(height, world, row_cost=None):
    """world + 1 row numbers cutting [0, height) into contiguous bands: equal heights, or -- with one cost per row -- equal
    cost (ptrs_plan_bands, the planner the C++ host's N-device render uses).  Deterministic: every rank that computes it
    from the same costs gets the same plan."""
    import ctypes as C
    import numpy as np
    from .integrator import load_library, _check
    b = (C.c_int32 * (int(world) + 1))()
    rc = None if row_cost is None else np.ascontiguousarray(row_cost, dtype=np.float32)
    if rc is not None and rc.shape[0] != int(height):
        raise ValueError("row_cost needs one entry per film row")
    _check(load_library().ptrs_plan_bands(int(height), int(world), C.c_void_p(rc.ctypes.data) if rc is not None else None, b))
    return [int(v) for v in b]


def probe_row_cost(pkg, camera, scene, max_depth, device=0, strips=0, spp=1, cache=True):
    """Per-row cost estimate for plan_bands: ONE render of the frame at `spp` samples per pixel with a device counter per sample row
    (ptrs_render_row_cost: every BVH query of a path is added to its row; no film), where round 3 made 64 strip renders (270 ms for
    Cornell against a 144 ms frame).  Ray counts are deterministic, so every rank gets the same numbers without talking to the
    others.  The result is cached per (scene, camera, resolution, depth, device): planning a second frame of the same view costs
    nothing.  `strips` is ignored (kept for callers of the old form)."""
    import ctypes as C
    import numpy as np
    from .integrator import PathIntegrator, SamplerBuilder, _check, _device_scene, abi, load_library
    cam = camera.to_abi()
    key = (bytes(cam), camera.film.width, camera.film.height, int(max_depth), int(device), int(spp))
    store = scene.__dict__.setdefault("_ptrs_row_cost", {})  # the cache lives and dies with the scene object
    if cache and key in store:
        return store[key].copy()
    integ = PathIntegrator(SamplerBuilder(spp, camera.film.get_sample_bounds()), max_depth, device=device)
    p = integ.params(camera)
    ds = _device_scene(scene, device)
    cost = np.zeros(camera.film.height, np.float32)
    st = abi.PtrsStats()
    _check(load_library().ptrs_render_row_cost(ds.handle, C.byref(cam), C.byref(p), C.c_void_p(cost.ctypes.data), C.byref(st)))
    if cache:
        store[key] = cost.copy()
    return cost


def plan_gain(height, world, row_cost):
    """What a cost-weighted plan is worth: (cost of the most expensive equal-height band) / (cost of the most expensive planned band).
    Callers keep equal bands when this is within noise of 1."""
    import numpy as np
    c = np.asarray(row_cost, dtype=np.float64)
    eq = [band_for_rank(height, r, world) for r in range(world)]
    pl = plan_bands(height, world, row_cost)
    worst = lambda bands: max(c[b:e].sum() for b, e in bands)
    return worst(eq) / max(worst([(pl[r], pl[r + 1]) for r in range(world)]), 1e-30)


_P2P_READY = set()  # process groups on which a collective has run before the first point-to-point batch (see gather_film_rows)


def gather_film_rows(film, height, rank, world, group=None, dst=0, bounds=None, mode="p2p", force=False):
    """film: (H, W, 4) tensor whose rows of this rank's band are valid (band_for_rank, or bounds[rank]:bounds[rank + 1] when
    a plan is given).  After the call rank `dst` holds the complete film.  `rank`, `world` and `dst` are ranks of `group`
    (the default group when None).
    mode "p2p" (default): every other rank sends exactly its band, `dst` receives each band straight into its rows of the
    film (one batch_isend_irecv: no padding, no staging copy -- bands of a cost-weighted plan differ in height).  Ranks
    with an empty band take no part in the batch, which RCCL only tolerates on a communicator that exists already: the
    first p2p call on a group therefore runs one dist.barrier(group) first (every rank of the group makes the same call,
    so every rank reaches it).  Peers are addressed by their GLOBAL rank (P2POp's convention), translated from the group's.
    mode "gather": one dist.gather of equal-size slabs, bands padded to the largest one (the round-2 form; also what
    `bench.py`'s world-size-1 collective smoke runs with force=True, where there is no peer to send to)."""
    if world == 1 and not force:
        return film
    band = (lambda r: (int(bounds[r]), int(bounds[r + 1]))) if bounds is not None else (lambda r: band_for_rank(height, r, world))
    b, e = band(rank)
    if mode == "p2p" and world > 1:
        key = id(group) if group is not None else None
        if key not in _P2P_READY:
            dist.barrier(group)
            _P2P_READY.add(key)
        peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
        ops = []
        if rank == dst:
            for r in range(world):
                rb, re = band(r)
                if r != dst and re > rb:
                    ops.append(dist.P2POp(dist.irecv, film[rb:re], peer(r), group))  # a contiguous slab of rows: received in place
        elif e > b:
            ops.append(dist.P2POp(dist.isend, film[b:e].contiguous(), peer(dst), group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return film
    gdst = dist.get_global_rank(group, dst) if group is not None else dst  # dist.gather's `dst` is a global rank as well
    max_rows = max(band(r)[1] - band(r)[0] for r in range(world))
    if e - b == max_rows:
        send = film[b:e]
    else:
        send = torch.zeros((max_rows,) + tuple(film.shape[1:]), dtype=film.dtype, device=film.device)
        send[: e - b] = film[b:e]
    send = send.contiguous()
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=gdst, group=group)
        for r in range(world):
            if r == dst:
                continue
            rb, re = band(r)
            film[rb:re] = bufs[r][: re - rb]
    else:
        dist.gather(send, None, dst=gdst, group=group)
    return film
