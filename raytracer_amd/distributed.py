"""Frame partitioning across the GPUs of one node: one process per GPU, torch.distributed over RCCL.

Sample split: rank r renders camera-sample indices [r*S, (r+1)*S) of every pixel into its own device accumulator
(sum rgb, n, sum of squares, -), then one sum-reduce of the accumulators to rank 0 — the only exchange per frame.
Tile split: rank r renders the 32x32 tiles r, r+world, r+2*world, ... (reference tileSize, raytracer.h:46); ranks own
disjoint pixels, so the same zero-initialised sum-reduce gathers the frame.
The scene, BVH, textures (and photon maps: same Philox seed => identical maps) are replicated on every GPU.
"""
import copy

import torch
import torch.distributed as dist

ACCUM_FLOATS_PER_PIXEL = 8


def alloc_accumulator(width, height, device):
    return torch.zeros(width * height * ACCUM_FLOATS_PER_PIXEL, dtype=torch.float32, device=device)


def sample_split_params(params, rank, spp_per_rank):
    p = copy.copy(params)
    p.sample_begin = rank * spp_per_rank
    p.sample_count = spp_per_rank
    if p.spp_max < p.sample_begin + p.sample_count or p.spp_min != p.spp_max:
        p.spp_min = p.spp_max = max(p.spp_max, p.sample_begin + p.sample_count)  # fixed spp: adaptive stop is per-pixel sequential
    return p


def tile_split_params(params, rank, world):
    p = copy.copy(params)
    p.tile_begin, p.tile_stride, p.tile_count = rank, world, 0
    return p


def render_partition(renderer, params, accum, group=None, dst=0):
    """Renders this rank's share into `accum` and sums all shares on rank `dst` (no-op reduce for one rank)."""
    renderer.render_accumulate(params, accum.data_ptr())
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum
