"""Pixel-range partition for one-process-per-GPU runs.

The reference's only partition key is the inclusive 1-based pixel range i1..i2 (read_inputs.f90:8-9;
restricting it reproduces exactly those pixels).  Pixel order is alpha-major, beta fastest
(geokerr_wrapper.f:258-259), so a contiguous range is a vertical strip of the image."""
import numpy as np


def strip(rank, world, npix):
    """Contiguous, near-equal i1..i2 (1-based inclusive) for `rank` of `world`; strips tile 1..npix."""
    base, rem = divmod(npix, world)
    lo = rank * base + min(rank, rem)
    n = base + (1 if rank < rem else 0)
    return lo + 1, lo + n


def ranges(rank, world, i1, i2, nphi, tile_cols=0):
    """Pixel ranges [(lo, hi)] (1-based inclusive) of `rank`: the share-out module pgrtrans uses (pgrtrans.f90:pixel_ranges).
    tile_cols <= 0: one contiguous near-equal strip of i1..i2.  tile_cols > 0: tiles of tile_cols image columns (a column =
    nphi consecutive pixels, alpha-major order) dealt round-robin over the ranks -- SURVEY 8e: captured and dead rays
    cluster in image regions, interleaving spreads them over the GPUs."""
    if tile_cols <= 0:
        lo, hi = strip(rank, world, i2 - i1 + 1)
        return [(i1 - 1 + lo, i1 - 1 + hi)] if hi >= lo else []
    w = tile_cols * nphi
    out = []
    for t in range((i1 - 1) // w, (i2 - 1) // w + 1):
        if t % world != rank:
            continue
        a, b = max(i1, t * w + 1), min(i2, (t + 1) * w)
        if b >= a:
            out.append((a, b))
    return out


def counts(world, npix, per_pixel):
    return [(strip(r, world, npix)[1] - strip(r, world, npix)[0] + 1) * per_pixel for r in range(world)]


def gather_tiles(tile, dist, dst=0):
    """Assemble the image on `dst` from per-rank tiles [ncams][npix_tile][nvals] (torch tensors,
    device = backend's: cuda for nccl/RCCL, cpu for gloo).  Ragged tiles are padded to the largest."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ncams, n, nv = tile.shape
    sizes = [torch.zeros(1, dtype=torch.int64, device=tile.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n], dtype=torch.int64, device=tile.device))
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = torch.zeros((ncams, nmax, nv), dtype=tile.dtype, device=tile.device)
    pad[:, :n] = tile
    outs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, outs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:, :s] for o, s in zip(outs, sizes)], dim=1)
