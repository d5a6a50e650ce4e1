"""Multi-GPU sharding of the film: one process per GPU, contiguous bands of
output rows per rank, one gather of the bands to rank 0 over RCCL (torch.distributed backend
"nccl") at the end of the render.  Each sample depends only on (pixel, sample index) -- the sampler
ignores the tile seed (sampler/sobol.rs:75-77) -- so ranks exchange nothing while tracing; a rank
re-traces the 2-row filter halo on each side of its band (film.rs:60-106) so that its rows are
complete and bit-identical to the single-GPU render."""
import torch
import torch.distributed as dist


def band_for_rank(height, rank, world):
    """Rows [begin, end) owned by `rank`: as even as possible, earlier ranks take the remainder."""
    base, rem = divmod(int(height), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_film_rows(film, height, rank, world, group=None, dst=0):
    """film: (H, W, 4) tensor whose rows band_for_rank(H, rank, world) are valid on this rank.
    After the call rank `dst` holds the complete film.  One collective (gather) of equal-size
    slabs; unequal bands are padded to the largest band."""
    if world == 1:
        return film
    max_rows = -(-int(height) // int(world))
    b, e = band_for_rank(height, rank, world)
    if e - b == max_rows:
        send = film[b:e]
    else:
        send = torch.zeros((max_rows,) + tuple(film.shape[1:]), dtype=film.dtype, device=film.device)
        send[: e - b] = film[b:e]
    send = send.contiguous()
    if rank == dst:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, bufs, dst=dst, group=group)
        for r in range(world):
            if r == dst:
                continue
            rb, re = band_for_rank(height, r, world)
            film[rb:re] = bufs[r][: re - rb]
    else:
        dist.gather(send, None, dst=dst, group=group)
    return film
