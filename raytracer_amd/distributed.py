"""Frame partitioning across the GPUs of one node: one process per GPU, torch.distributed over RCCL.

Sample split: rank r renders camera-sample indices [r*S, (r+1)*S) of every pixel into its own device accumulator
(sum rgb, n, sum of squares, -), then one sum-reduce of the accumulators to rank 0 — the only exchange per frame.
Tile split: rank r renders the 32x32 tiles r, r+world, r+2*world, ... (reference tileSize, raytracer.h:46); ranks own
disjoint pixels, so the same zero-initialised sum-reduce gathers the frame.
Hybrid (BASELINE config 5: 8 GPUs as 2 tile groups x 4 sample ranges): rank r renders the tiles of tile group r % T with the
sample range of group r // T; the one sum-reduce still assembles the frame, because every (pixel, sample) is rendered by
exactly one rank.
The scene, BVH, textures (and photon maps: same Philox seed => identical maps) are replicated on every GPU.

Stream order: the renderer core runs on its own HIP stream and every C-ABI call blocks until its work has finished, so
what it wrote is visible to torch afterwards; in the other direction (torch zeroing / unpacking an accumulator that the
next rt_render_accumulate or rt_resolve reads) the helpers here synchronise torch's current stream before they return.
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


def hybrid_split_params(params, rank, tile_groups, sample_groups, spp_per_group):
    """rank -> (tile group rank % tile_groups, sample group rank // tile_groups): tiles t = tg, tg + T, ... (raytracer.h:46)
    and camera samples [sg * S, (sg + 1) * S).  tile_groups * sample_groups must be the world size."""
    if rank < 0 or rank >= tile_groups * sample_groups:
        raise ValueError("rank outside the tile_groups x sample_groups grid")
    p = sample_split_params(params, rank // tile_groups, spp_per_group)
    p.tile_begin, p.tile_stride, p.tile_count = rank % tile_groups, tile_groups, 0
    if p.spp_max < sample_groups * spp_per_group:
        p.spp_min = p.spp_max = sample_groups * spp_per_group
    return p


def _sync_torch_stream(t):
    """torch work on `t` is finished before the renderer core's own stream touches it."""
    if t.is_cuda:
        torch.cuda.current_stream(t.device).synchronize()


class PendingReduce:
    """An accumulator exchange in flight (RCCL runs it on its own stream); wait() completes it on `dst`."""

    def __init__(self, accum, packed, work, is_dst):
        self.accum, self.packed, self.work, self.is_dst = accum, packed, work, is_dst

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.packed is not None and self.is_dst:
            self.accum.view(-1, ACCUM_FLOATS_PER_PIXEL)[:, :4] = self.packed.view(-1, 4)
            _sync_torch_stream(self.accum)   # the unpacked sums are in place before rt_resolve / the next render reads them
        self.packed = None
        return self.accum


def reduce_accumulators(accum, group=None, dst=0, sums_only=True, async_op=False):
    """Sums the ranks' accumulators on rank `dst`.  sums_only: exchange (sum rgb, n) only — half the bytes; the sums of
    squares and the done flag matter to adaptive sampling alone, which partitioned rendering does not use (fixed spp)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return PendingReduce(accum, None, None, True)
    is_dst = dist.get_rank(group) == dst
    if accum.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal of the multi-rank path on a box without one GPU per rank (gloo does not reduce device tensors):
        # staged through the host, blocking
        host = (accum.view(-1, ACCUM_FLOATS_PER_PIXEL)[:, :4].contiguous() if sums_only else accum).cpu()
        dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
        if is_dst:
            if sums_only:
                accum.view(-1, ACCUM_FLOATS_PER_PIXEL)[:, :4] = host.to(accum.device)
            else:
                accum.copy_(host)
        return PendingReduce(accum, None, None, is_dst)
    if sums_only:
        packed = accum.view(-1, ACCUM_FLOATS_PER_PIXEL)[:, :4].contiguous()
        work = dist.reduce(packed, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending = PendingReduce(accum, packed, work, is_dst)
    else:
        work = dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending = PendingReduce(accum, None, work, is_dst)
    if not async_op:
        pending.wait()
    return pending


def render_partition(renderer, params, accum, group=None, dst=0, sums_only=True, async_op=False):
    """Renders this rank's share into `accum` (blocking), then sums all shares on rank `dst` (no exchange for one rank).
    async_op: returns with the exchange still in flight so that the next frame's rendering overlaps it; the caller
    keeps `accum` untouched until PendingReduce.wait()."""
    _sync_torch_stream(accum)   # (the caller's accum.zero_() has landed)
    renderer.render_accumulate(params, accum.data_ptr())
    return reduce_accumulators(accum, group=group, dst=dst, sums_only=sums_only, async_op=async_op)
