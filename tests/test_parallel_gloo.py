"""Multi-rank band sharding + gather (pathtracer-rs_amd/parallel.py) rehearsed on the CPU with the
gloo backend, world_size 2 and 3: each rank renders its band of rows (here with the oracle, since
the product has no CPU path) and rank 0 must end up with exactly the single-process film."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import CORNELL, ROOT


def _worker(rank, world, port, w, h, spp, depth, out_path, bounds=None, mode="p2p"):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("pathtracer-rs_amd")
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    from oracle import orc
    cam, scene = pkg.import_scene(CORNELL, (w, h))
    b, e = (bounds[rank], bounds[rank + 1]) if bounds else par.band_for_rank(h, rank, world)
    film_np, _, _ = orc.OracleScene(scene).render(cam, orc.make_params(w, h, spp, depth, row_begin=b, row_end=e), n_threads=1)
    film = torch.from_numpy(np.concatenate([film_np["rgb"], film_np["weight"][..., None]], axis=-1).copy())
    par.gather_film_rows(film, h, rank, world, bounds=bounds, mode=mode)
    if rank == 0:
        np.save(out_path, film.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,bounds,mode", [(2, None, "p2p"), (3, None, "p2p"), (2, [0, 5, 22], "p2p"), (3, [0, 9, 10, 22], "p2p"), (3, [0, 9, 9, 22], "p2p"),
                                               (2, [0, 5, 22], "gather"), (3, None, "gather")])
def test_band_gather_equals_single_process(tmp_path, world, bounds, mode):
    """Equal bands and planned (unequal, also empty) bands, exact-size point-to-point transfers and the padded gather: the
    gathered film is the single-process film, bit for bit."""
    w, h, spp, depth = 24, 22, 2, 3
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, spp, depth, out, bounds, mode), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import importlib
    pkg = importlib.import_module("pathtracer-rs_amd")
    from oracle import orc
    cam, scene = pkg.import_scene(CORNELL, (w, h))
    ref, _, _ = orc.OracleScene(scene).render(cam, orc.make_params(w, h, spp, depth), n_threads=1)
    got = np.load(out)
    assert np.array_equal(got[..., :3].view(np.uint32), ref["rgb"].view(np.uint32))
    assert np.array_equal(got[..., 3].view(np.uint32), ref["weight"].view(np.uint32))


def _smoke_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    film = torch.arange(64 * 8 * 4, dtype=torch.float32).reshape(64, 8, 4)
    ref = film.clone()
    par.gather_film_rows(film, 64, 0, 1, bounds=[0, 64], mode="gather", force=True)
    assert torch.equal(film, ref)
    dist.destroy_process_group()


def test_world_size_one_gather_smoke():
    """What bench.py's collective smoke runs on the GPU box with the nccl backend, rehearsed with gloo: a process group of one rank,
    the padded gather forced through."""
    mp.spawn(_smoke_worker, args=(1, _free_port()), nprocs=1, join=True)


def test_band_partition_covers_rows():
    import importlib
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    for h in (1, 7, 1024, 2160):
        for world in (1, 2, 3, 8):
            bands = [par.band_for_rank(h, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in bands) - min(e - b for b, e in bands) <= 1


def test_plan_bands_equal_cost():
    """parallel.plan_bands (ptrs_plan_bands): contiguous, covering, and with per-row costs no band above 1.25 x the mean cost
    where equal heights would give 1.6 x."""
    import importlib
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    assert par.plan_bands(10, 3) == [0, 4, 7, 10]
    h, world = 1024, 8
    cost = 1.0 + 4.0 * np.exp(-((np.arange(h) - 300.0) / 120.0) ** 2)  # a bump of expensive rows
    b = par.plan_bands(h, world, cost)
    assert b[0] == 0 and b[-1] == h and all(b[i] < b[i + 1] for i in range(world))
    per = np.array([cost[b[i]:b[i + 1]].sum() for i in range(world)])
    even = np.array([cost[h * i // world:h * (i + 1) // world].sum() for i in range(world)])
    assert per.max() / per.mean() < 1.25 < even.max() / even.mean()


def _subgroup_worker(rank, world, port, mode):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("pathtracer-rs_amd.parallel")
    grp = dist.new_group([1, 2])  # group ranks 0, 1 = global ranks 1, 2
    if rank in (1, 2):
        g_rank = rank - 1
        want = torch.arange(20 * 4 * 4, dtype=torch.float32).reshape(20, 4, 4)
        bounds = [0, 7, 20]
        film = torch.zeros_like(want)
        film[bounds[g_rank]:bounds[g_rank + 1]] = want[bounds[g_rank]:bounds[g_rank + 1]]
        par.gather_film_rows(film, 20, g_rank, 2, group=grp, dst=0, bounds=bounds, mode=mode)
        if g_rank == 0:
            assert torch.equal(film, want)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["p2p", "gather"])
def test_gather_on_a_subgroup_uses_group_ranks(mode):
    """rank / world / dst are ranks of `group`; the peers of the point-to-point batch (and dist.gather's dst) are global ranks"""
    mp.spawn(_subgroup_worker, args=(3, _free_port(), mode), nprocs=3, join=True)
