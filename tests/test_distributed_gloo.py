"""N > 1 path on CPU: two processes over gloo run raytracer_amd.distributed exactly as bench.py does over RCCL —
sample-range split and tile split, accumulators summed on rank 0 — with the host-emulation build standing in for the GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path

W, H, SPP = 64, 48, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_ffi
    from common import params
    from raytracer_amd import Scene, distributed
    scene = Scene.load(scene_path("cornell_box.xml"))
    r = emu_ffi.EmuRenderer()
    r.LoadScene(scene)
    base = params(W, H, spp=SPP * world, jitter=1, bounces=2)
    if mode == "hybrid":   # BASELINE config 5's partition in small: 2 tile groups x 2 sample ranges
        base = params(W, H, spp=SPP * 2, jitter=1, bounces=2)
        p = distributed.hybrid_split_params(base, rank, 2, world // 2, SPP)
    elif mode.startswith("sample"):
        p = distributed.sample_split_params(base, rank, SPP)
    else:
        p = distributed.tile_split_params(base, rank, world)
    if mode == "sample_pipelined":
        # bench.py's N > 1 loop: two accumulators, each frame's exchange left in flight while the next frame renders
        accs = [distributed.alloc_accumulator(W, H, "cpu") for _ in range(2)]
        pend = [None, None]
        for frame in range(3):
            k = frame % 2
            if pend[k] is not None:
                pend[k].wait()
            accs[k].zero_()
            pend[k] = distributed.render_partition(r, p, accs[k], async_op=True)
        for q in pend:
            q.wait()
        acc = accs[0]   # frame 2
    else:
        acc = distributed.alloc_accumulator(W, H, "cpu")
        distributed.render_partition(r, p, acc, sums_only=(mode != "sample_full"))
    dist.barrier()
    if rank == 0:
        img = r.resolve(base, acc.data_ptr())
        full = r.Render(base)
        np.savez(out_path, split=img.radiance, full=full.radiance, split8=img.img, full8=full.img)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sample", "sample_full", "sample_pipelined", "tile"])
def test_two_rank_partition_equals_single_render(tmp_path, mode):
    out = str(tmp_path / f"{mode}.npz")
    mp.spawn(_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
    d = np.load(out)
    assert (d["split"][..., 3] == SPP * 2).all()
    if mode == "tile":
        assert np.array_equal(d["split"], d["full"]) and np.array_equal(d["split8"], d["full8"])
    else:
        assert np.allclose(d["split"], d["full"], rtol=1e-6, atol=1e-7)
        assert np.abs(d["split8"].astype(int) - d["full8"].astype(int)).max() <= 1


def test_four_rank_hybrid_partition_equals_single_render(tmp_path):
    """SURVEY 8e hybrid partition (C5: 2 tile groups x 4 sample ranges on 8 GPUs), here 2 x 2 on four gloo ranks: every
    (pixel, sample) is rendered by exactly one rank and one sum-reduce assembles the frame."""
    out = str(tmp_path / "hybrid.npz")
    mp.spawn(_worker, args=(4, _free_port(), "hybrid", out), nprocs=4, join=True)
    d = np.load(out)
    assert (d["split"][..., 3] == SPP * 2).all()
    assert np.allclose(d["split"], d["full"], rtol=1e-6, atol=1e-7)
    assert np.abs(d["split8"].astype(int) - d["full8"].astype(int)).max() <= 1
