#!/usr/bin/env python3
"""bench.py — headline benchmark of the renderer core on MI355X.

Workload (BASELINE.json configs[1], "C2"): scenes/c2_mesh_room.xml — the reference's scene_5.xml layout with two
seeded procedural triangle meshes (~81k triangles), reflection + refraction, 3 bounces, hard point light, photon
term off — at 1920x1080, 32 camera samples per pixel per step, Philox seed 1.

One step = one frame's worth of samples through the whole hot path on every rank:
rt_render_accumulate (raygen -> [trace_closest -> shade -> shadow]*4 levels -> accumulate) into a device
accumulator that already lives in HBM, then (N > 1) the sample-split exchange: an RCCL sum-reduce of the
accumulators to rank 0.  Rank r renders sample indices [32 r, 32 r + 32), so per-GPU work is fixed as N grows
(weak scaling).  value = rays traced by all ranks per step / step time (max over ranks).

1 ray = one closest-hit or any-hit query; the per-step ray count is measured once, untimed, with the
counted kernel variants (the workload is deterministic, so the timed steps trace exactly the same rays).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPP_PER_STEP = 32
WIDTH, HEIGHT = 1920, 1080
SCENE = os.path.join(ROOT, "scenes", "c2_mesh_room.xml")
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3    # FP32 vector peak


def make_params(rank):
    from raytracer_amd import default_params
    p = default_params()
    p.width, p.height = WIDTH, HEIGHT
    p.spp_min = p.spp_max = 65536  # fixed sample count; the range below selects what this call renders
    p.sample_begin = rank * SPP_PER_STEP
    p.sample_count = SPP_PER_STEP
    p.pixel_jitter = 1
    p.max_bounces = 3
    p.photons = 0
    p.seed_lo = 1
    return p


def cpu_baseline(scene, target_seconds=12.0):
    """The CPU oracle (reference-structured restatement: Θ(nodes) scene-graph walk, BVH visiting both children, std::thread
    over 32x32 tiles) timed on this host on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    cores = min(16, os.cpu_count() or 1)
    orc = oracle_ffi.OracleScene(scene)
    p = make_params(0)
    p.sample_count = 1
    t0 = time.perf_counter()
    _, cnt = orc.render(p, cores, want_counters=True)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(8 * SPP_PER_STEP, target_seconds / max(t1, 1e-3))))
    p.sample_count = spp
    t0 = time.perf_counter()
    _, cnt = orc.render(p, cores, want_counters=True)
    dt = time.perf_counter() - t0
    rays = cnt["rays_primary"] + cnt["rays_secondary"] + cnt["rays_shadow"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{WIDTH}x{HEIGHT} x {spp} spp of the same scene/params, {rays} rays in {dt:.2f} s, oracle/liboracle.so "
                      f"(reference structure: no TLAS, BVH visits both children), {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from raytracer_amd import Renderer, Scene, distributed
    scene = Scene.load(SCENE)
    r = Renderer(local_rank)
    r.LoadScene(scene)
    p = make_params(rank)
    accum = torch.zeros(WIDTH * HEIGHT * 8, dtype=torch.float32, device=f"cuda:{local_rank}")

    def step():
        accum.zero_()
        torch.cuda.synchronize()
        # blocking render of this rank's sample range into `accum`, then the sample-split exchange:
        # one RCCL sum-reduce of the accumulators to rank 0 over xGMI
        distributed.render_partition(r, p, accum)

    # one counted, untimed step: algorithmic work per step
    r.counters(enable=True, reset=True)
    step()
    torch.cuda.synchronize()
    cnt = r.counters()
    r.counters(enable=False)
    rays_per_step = cnt["rays_primary"] + cnt["rays_secondary"] + cnt["rays_shadow"]

    for _ in range(args.warmup):
        step()
    r.profile(enable=True, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = r.profile()
    r.profile(enable=False)

    t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
    rays = torch.tensor([float(rays_per_step)], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    total_rays_per_step = float(rays.item())

    if rank == 0:
        # final image of the last step (outside the timed region): sanity that the accumulators hold a frame
        pr = make_params(0)
        img = r.resolve(pr, accum.data_ptr())
        n_expected = SPP_PER_STEP * world
        assert abs(float(img.radiance[..., 3].mean()) - n_expected) < 1e-3, "accumulator does not hold the expected samples"

        tc = prof["trace_closest"]
        # SURVEY §8d algorithmic bytes of the closest-hit kernel: 48 B per ray (32 B ray in, 16 B hit out) + 32 B per BVH box
        # tested (two per 64-byte node fetch) + 48 B per triangle slot + 96 B per instance transform pair.
        tc_rays = cnt["rays_primary"] + cnt["rays_secondary"]
        tc_bytes = 48.0 * tc_rays + 32.0 * cnt["closest_node_visits"] + 48.0 * cnt["closest_tri_tests"] + 96.0 * cnt["closest_inst_tests"]
        tc_flops = 36.0 * cnt["closest_inst_tests"] + 24.0 * cnt["closest_node_visits"] + 51.0 * cnt["closest_tri_tests"]
        launches_per_step = tc["launches"] / max(args.steps, 1)
        avg_ms = tc["ms"] / max(tc["launches"], 1)
        achieved = (tc_bytes / max(launches_per_step, 1)) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_trace_closest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        kernel_ms = {k: round(v["ms"] / max(args.steps, 1), 3) for k, v in prof.items() if v["launches"]}
        out = {
            "metric": "Mrays/s", "value": total_rays_per_step * args.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: scenes/c2_mesh_room.xml (scene_5 layout, 2 procedural meshes, 81k triangles), "
                                   f"{WIDTH}x{HEIGHT}, {SPP_PER_STEP} spp per GPU per step, 3 bounces reflection+refraction, "
                                   "hard point-light shadows, photons off, Philox seed 1",
                       "rays_per_step_per_gpu": rays_per_step, "paths_per_step_per_gpu": cnt["paths"],
                       "mpaths_per_s": cnt["paths"] * world * args.steps / dt / 1e6,
                       "partition": "sample-range split, RCCL sum-reduce of the accumulator" if world > 1 else "single GPU",
                       "kernel_ms_per_step": kernel_ms},
            "roofline": {"bound": "hbm", "kernel": "k_trace_closest", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": tc_bytes / max(launches_per_step, 1),
                         "avg_launch_ms": avg_ms, "launches_per_step": launches_per_step,
                         "valu_tflops": (tc_flops / max(launches_per_step, 1)) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                         "valu_frac": ((tc_flops / max(launches_per_step, 1)) / (avg_ms * 1e-3) / 1e12) / VALU_PEAK_TFLOPS if avg_ms > 0 else 0.0,
                         "note": "scene working set (~9 MB of BVH nodes + triangle slots) is cache resident: the kernel is "
                                 "latency/VALU bound; HBM fraction is reported as the contract asks, the FP32 VALU fraction beside it"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
