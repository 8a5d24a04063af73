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

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3    # FP32 vector peak


class Config:
    """One BASELINE.json configuration.  The driver runs the default (c2, configs[1]: the largest that BASELINE quotes
    on ONE MI355X); the others are selectable with --config for the measurements recorded in BASELINE.md / DESIGN.md."""

    def __init__(self, name, workload, width, height, spp_per_step, scene, tweak, cpu_tile_stride=1, progressive_spp=0):
        self.name, self.workload, self.width, self.height = name, workload, width, height
        self.spp_per_step, self.scene_fn, self.tweak = spp_per_step, scene, tweak
        self.cpu_tile_stride = cpu_tile_stride   # the CPU baseline renders every n-th 32x32 tile (bounded sample)
        # > 0: consecutive steps render consecutive sample ranges of a frame of this many spp, as the passes of a
        # progressive render do (C4: the renderer keeps per-pixel photon-lookup hints between passes, so repeating ONE
        # range would hand every lookup its own previous answer)
        self.progressive_spp = progressive_spp

    def sample_begin(self, step, rank, world):
        if self.progressive_spp <= 0:
            return rank * self.spp_per_step
        return ((step * world + rank) * self.spp_per_step) % self.progressive_spp

    def scene(self):
        return self.scene_fn()

    def params(self, rank):
        from raytracer_amd import default_params
        p = default_params()
        p.width, p.height = self.width, self.height
        p.spp_min = p.spp_max = 65536  # fixed sample count; the range below selects what this call renders
        p.sample_begin = rank * self.spp_per_step
        p.sample_count = self.spp_per_step
        p.pixel_jitter = 1
        p.max_bounces = 3
        p.photons = 0
        p.seed_lo = 1
        self.tweak(p)
        return p


HOST_BVH = True   # --device-bvh: meshes are handed over without a tree and rt_scene_commit builds it on the device


def _load(path):
    def f():
        from raytracer_amd import Scene
        return Scene.load(os.path.join(ROOT, path), host_bvh=HOST_BVH)
    return f


def _stress():
    from raytracer_amd import Scene
    return Scene.stress(seed=7, n_instances=2048, subdiv=4, ground_res=1024, width=3840, height=2160, host_bvh=HOST_BVH)


def sphere_field_xml(n=2944, seed=11, width=1920, height=1080):
    """A seeded stand-in for the reference's scenes/volumetric_spiral.xml (2 944 glass spheres under one group node + a
    floor sphere, one size-0 point light): the scene-graph stress case, Theta(nodes) per ray in the reference."""
    import numpy as np
    rng = np.random.default_rng(seed)
    mats = ["glassClear", "glassRed", "glassBlue", "glassGold"]
    out = ['<xml><scene>', '<object type="sphere" name="floor" material="floorMat"><scale value="200"/><translate z="-200"/></object>',
           '<object name="helix">']
    for i in range(n):
        t = i / max(n - 1, 1)
        ang = 40.0 * t * 2.0 * np.pi
        rad = 12.0 + 30.0 * t + rng.normal() * 3.0
        x, y, z = rad * np.cos(ang) + rng.normal() * 2.0, 30.0 + 320.0 * t + rad * np.sin(ang) * 0.3, 1.0 + 40.0 * t + abs(rng.normal()) * 6.0
        sc = 1.5 + 3.0 * rng.random()
        out.append(f'<object type="sphere" name="s{i}" material="{mats[int(rng.integers(0, 4))]}"><scale value="{sc:.4f}"/>'
                   f'<translate x="{x:.4f}" y="{y:.4f}" z="{z:.4f}"/></object>')
    out.append('</object>')
    out.append('<material type="blinn" name="floorMat"><diffuse r="0.35" g="0.35" b="0.4"/><specular value="0.2"/><glossiness value="30"/></material>')
    for name, kd, ab in (("glassClear", "0.05 0.05 0.05", "0.002 0.002 0.002"), ("glassRed", "0.12 0.03 0.03", "0.002 0.01 0.01"),
                         ("glassBlue", "0.03 0.05 0.12", "0.01 0.008 0.002"), ("glassGold", "0.12 0.09 0.03", "0.01 0.008 0.002")):
        r, g, b = kd.split()
        ar, ag, ab_ = ab.split()
        out.append(f'<material type="blinn" name="{name}"><diffuse r="{r}" g="{g}" b="{b}"/><specular value="0.9"/><glossiness value="95"/>'
                   f'<refraction value="1.0" index="1.52"/><reflection value="0.3"/><absorption r="{ar}" g="{ag}" b="{ab_}"/></material>')
    out.append('<light type="ambient"><intensity value="0.12"/></light>')
    out.append('<light type="point"><intensity value="1.1"/><position x="0" y="0" z="444"/></light>')
    out.append(f'</scene><camera><position x="0" y="-25" z="12"/><target x="0" y="200" z="30"/><up x="0" y="0" z="1"/><fov value="55"/>'
               f'<width value="{width}"/><height value="{height}"/></camera></xml>')
    return "\n".join(out)


def _sphere_field():
    from raytracer_amd import Scene
    return Scene.from_string(sphere_field_xml())


def _c1(p):
    p.pixel_jitter, p.max_bounces, p.light_size_scale = 0, 0, 0.0


def _c3(p):
    p.mc_bounces = 2


def _c4(p):
    p.photons = 1


CONFIGS = {
    "c1": Config("c1", "C1: spheres + planes scene (tests/golden/scenes/spheres_room.xml), Blinn + hard point-light shadows, "
                 "pixel centre, 0 bounces", 800, 600, 1, _load("tests/golden/scenes/spheres_room.xml"), _c1),
    "c2": Config("c2", "C2: scenes/c2_mesh_room.xml (scene_5 layout, 2 procedural meshes, 81k triangles), 3 bounces "
                 "reflection+refraction, hard point-light shadows, photons off", 1920, 1080, 32, _load("scenes/c2_mesh_room.xml"),
                 lambda p: None),
    "c3": Config("c3", "C3: scenes/c3_cornell_textured.xml (Cornell box, checker floor, brick wall, cloud environment), Monte-Carlo "
                 "indirect diffuse 2 bounces + 3 specular bounces, soft shadows 16..128 rays; 8 of the 1024 spp per step", 3840, 2160, 8,
                 _load("scenes/c3_cornell_textured.xml"), _c3, cpu_tile_stride=16),
    "c4": Config("c4", "C4: Cornell box (tests/golden/scenes/cornell_box.xml), 2 x 100k photons rebuilt every step, k=128 r=3 gather, "
                 "soft shadows, 3 bounces; step i renders samples [8i, 8i+8) of a 128-spp frame", 1920, 1080, 8,
                 _load("tests/golden/scenes/cornell_box.xml"), _c4, cpu_tile_stride=8, progressive_spp=128),
    "spheres": Config("spheres", "scene-graph stress (stand-in for the reference's scenes/volumetric_spiral.xml): 2944 glass spheres under "
                      "one group node + floor sphere, point light of size 0, 3 bounces reflection + refraction", 1920, 1080, 4, _sphere_field,
                      lambda p: None, cpu_tile_stride=16),
    "finalproject": Config("finalproject", "the scene the reference's main.cpp:12 hard-codes (scenes/finalProject.xml as the data fixture "
                           "tests/golden/scenes/ref_finalProject.xml: 5 walls, 64 glass spheres in 4 rotated rings, a procedural teapot for the "
                           "absent teapot.obj, sphere lights of size 4 and 8), 3 bounces, soft shadows 16..128 rays, photons off", 1920, 1080, 8,
                           _load("tests/golden/scenes/ref_finalProject.xml"), lambda p: None, cpu_tile_stride=16),
    "c5": Config("c5", "C5: synthetic stress scene, 12.6 M triangles in one BVH (2048 icospheres + displaced ground), 2 sphere "
                 "lights, 3 bounces; 4 of the 4096 spp per step", 3840, 2160, 4, _stress, lambda p: None, cpu_tile_stride=509),
}


def kernel_source_sha():
    """Hash of the kernel sources: PMC summaries under profiles/ are valid for exactly the build they were taken from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "raytracer_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc_summary(config_name, src_sha):
    """The newest profiles/r*_pmc_summary_<config>.json whose _meta.kernel_source_sha matches this build, else None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_summary_{config_name}.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_meta", {}).get("kernel_source_sha") == src_sha:
            d["_meta"]["file"] = os.path.relpath(f, ROOT)
            best = d
    return best


def cpu_baseline(cfg, scene, renderer, target_seconds=12.0):
    """The CPU oracle (reference-structured restatement: Θ(nodes) scene-graph walk, BVH visiting both children, std::thread
    over 32x32 tiles) timed on this host on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi
    # threads: the CPU share of a one-GPU box is 16 cores (os.cpu_count() reports the whole machine there); BENCH_CPU_THREADS
    # overrides.  Both numbers are printed.
    nproc = os.cpu_count() or 1
    try:
        nproc_affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        nproc_affinity = nproc
    cores = int(os.environ.get("BENCH_CPU_THREADS", "0")) or min(16, nproc_affinity)
    if not HOST_BVH:   # the oracle walks the trees the device built
        scene = scene.with_bvh({i: renderer.GetMeshBVH(i) for i in range(scene.desc.n_meshes)})
    orc = oracle_ffi.OracleScene(scene)
    WIDTH, HEIGHT, SPP_PER_STEP = cfg.width, cfg.height, cfg.spp_per_step
    p = cfg.params(0)
    p.tile_begin, p.tile_stride, p.tile_count = 0, cfg.cpu_tile_stride, 0
    if p.photons:
        orc.photons_build(p)   # (outside the timed region: the GPU side's ray count excludes photon rays as well)
    p.sample_count = 1
    t0 = time.perf_counter()
    cpu_img, cnt = orc.render(p, cores, want_counters=True)
    t1 = time.perf_counter() - t0
    spp, dt = 1, t1
    if t1 < 0.5 * target_seconds:
        spp = int(max(1, min(8 * SPP_PER_STEP, target_seconds / max(t1, 1e-3))))
        p.sample_count = spp
        t0 = time.perf_counter()
        cpu_img, cnt = orc.render(p, cores, want_counters=True)
        dt = time.perf_counter() - t0
    # the same pixels and samples on the GPU: PSNR of the 8-bit images and mean radiance difference (BASELINE metric)
    import numpy as np
    if p.photons:   # the comparison isolates rendering: the oracle gathers from the device's photons
        renderer.BuildPhotonMaps(p)
        orc.photons_set_tree(0, renderer.GetPhotonTree(0))
        orc.photons_set_tree(1, renderer.GetPhotonTree(1))
        cpu_img = orc.render(p, cores)
    g = renderer.Render(p)
    covered = cpu_img.sample_count > 0
    d = g.img[covered].astype(np.float64) - cpu_img.img[covered].astype(np.float64)
    mse = float((d * d).mean())
    psnr = 99.0 if mse == 0 else 10.0 * float(np.log10(255.0 * 255.0 / mse))
    mean_abs = float(np.abs(g.radiance[covered][:, :3] - cpu_img.radiance[covered][:, :3]).mean())
    rays = cnt["rays_primary"] + cnt["rays_secondary"] + cnt["rays_shadow"]
    tiles = "all 32x32 tiles" if cfg.cpu_tile_stride == 1 else f"every {cfg.cpu_tile_stride}th 32x32 tile"
    # second row: the same CPU renderer with the GPU's data structures — an instance BVH over the primitive nodes and
    # the mesh BVHs, walked front to back and pruned — so that the GPU / CPU ratio is not inflated by the reference's
    # visit-everything recursion (Theta(nodes) per ray in sphere-heavy scenes)
    orc.set_accel(1)
    q = cfg.params(0)
    q.tile_begin, q.tile_stride, q.tile_count = 0, cfg.cpu_tile_stride, 0
    q.sample_count = spp
    t0 = time.perf_counter()
    orc.render(q, cores)
    dt_fast = time.perf_counter() - t0
    orc.set_accel(0)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "nproc": nproc, "nproc_affinity": nproc_affinity,
            "ordered_pruned_bvh": {"value": rays / dt_fast / 1e6, "unit": "Mrays/s", "cores": cores,
                                   "note": "same oracle with the GPU's data structures: instance BVH over the primitive nodes' world boxes + mesh BVHs, "
                                           "both walked front to back with pruning (timing row only)"},
            "psnr_gpu_vs_cpu_db": psnr, "mean_abs_radiance_diff": mean_abs, "pixels_compared": int(covered.sum()),
            "sample": f"{WIDTH}x{HEIGHT}, {tiles}, {spp} spp of the same scene/params, {rays} rays in {dt:.2f} s, oracle/liboracle.so "
                      f"(reference structure: no TLAS, BVH visits both children, kd-tree + heap photon gather), {cores} threads"}


def main():
    global HOST_BVH
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="default run (c2, one GPU): skip the short runs of the other configurations")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--soft-clear", type=int, default=1, help="0: trace every soft-shadow group (no clear-view / full-shadow proofs)")
    ap.add_argument("--device-bvh", action="store_true", help="mesh BVHs built at rt_scene_commit on the device instead of by the host loader")
    args = ap.parse_args()
    HOST_BVH = not args.device_bvh
    cfg = CONFIGS[args.config]
    WIDTH, HEIGHT, SPP_PER_STEP = cfg.width, cfg.height, cfg.spp_per_step

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # BENCH_REHEARSE_GLOO=1: every rank on GPU 0 with the gloo backend — a rehearsal of the N > 1 plumbing on a one-GPU
    # box (its number means nothing: the ranks share the card and the exchange is staged through the host)
    rehearse = world > 1 and os.environ.get("BENCH_REHEARSE_GLOO") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    stat_dev = "cpu" if rehearse else f"cuda:{local_rank}"

    out = run_config(cfg, args, rank, local_rank, world, stat_dev, dist, brief=False)
    if rank == 0:
        if world == 1 and args.config == "c2" and not args.no_other_configs:
            # every other BASELINE configuration (and the scene-graph stress case) in the same driver-run line: short runs
            # (1 warm-up + 3 steps each), compact rows; their long CPU rows stay in profiles/ and BASELINE.md
            out["config"]["other_configs"] = {}
            for name in ("c1", "c3", "c4", "c5", "spheres", "finalproject"):
                short = argparse.Namespace(**vars(args))
                short.steps, short.warmup = 3, 1
                t0 = time.perf_counter()
                row = run_config(CONFIGS[name], short, rank, local_rank, world, stat_dev, dist, brief=True)
                row["wall_s"] = round(time.perf_counter() - t0, 2)
                out["config"]["other_configs"][name] = row
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def psnr_vs_oracle(cfg, scene, renderer, stride):
    """GPU against the CPU oracle on every `stride`-th 32x32 tile, one sample per pixel of the configuration's own integrator
    settings at the same Philox seed: PSNR of the 8-bit images and mean radiance difference (bounded: a second or two of CPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_ffi
    if not HOST_BVH:
        scene = scene.with_bvh({i: renderer.GetMeshBVH(i) for i in range(scene.desc.n_meshes)})
    orc = oracle_ffi.OracleScene(scene)
    p = cfg.params(0)
    p.tile_begin, p.tile_stride, p.tile_count = 0, stride, 0
    p.sample_count = 1
    if p.photons:
        renderer.BuildPhotonMaps(p)
        orc.photons_set_tree(0, renderer.GetPhotonTree(0))
        orc.photons_set_tree(1, renderer.GetPhotonTree(1))
    t0 = time.perf_counter()
    c = orc.render(p, min(16, len(os.sched_getaffinity(0))))
    cpu_s = time.perf_counter() - t0
    g = renderer.Render(p)
    covered = c.sample_count > 0
    d = g.img[covered].astype(np.float64) - c.img[covered].astype(np.float64)
    mse = float((d * d).mean())
    return {"psnr_db": 99.0 if mse == 0 else 10.0 * float(np.log10(255.0 * 255.0 / mse)),
            "mean_abs_radiance_diff": float(np.abs(g.radiance[covered][:, :3] - c.radiance[covered][:, :3]).mean()),
            "pixels": int(covered.sum()), "tile_stride": stride, "spp": 1, "cpu_s": round(cpu_s, 2)}


def run_config(cfg, args, rank, local_rank, world, stat_dev, dist, brief):
    """Loads, commits and times one configuration.  brief: a compact row for the main line's other_configs."""
    import torch
    WIDTH, HEIGHT, SPP_PER_STEP = cfg.width, cfg.height, cfg.spp_per_step
    from raytracer_amd import Renderer, Scene, distributed
    t0 = time.perf_counter()
    scene = cfg.scene()       # host: XML / generator, meshes, BVH build
    scene_load_s = time.perf_counter() - t0
    r = Renderer(local_rank)
    r.set_option("soft_clear", args.soft_clear)
    t0 = time.perf_counter()
    r.LoadScene(scene)        # rt_scene_commit: device layout + upload
    scene_commit_s = time.perf_counter() - t0
    p = cfg.params(rank)
    # two frame accumulators: the RCCL exchange of one frame overlaps the rendering of the next (N > 1)
    accums = [torch.zeros(WIDTH * HEIGHT * 8, dtype=torch.float32, device=f"cuda:{local_rank}") for _ in range(2 if world > 1 else 1)]
    pending = [None] * len(accums)
    frame = [0]

    def step():
        k = frame[0] % len(accums)
        frame[0] += 1
        if pending[k] is not None:
            pending[k].wait()
        p.sample_begin = cfg.sample_begin(frame[0] - 1, rank, world)
        accum = accums[k]
        accum.zero_()
        torch.cuda.synchronize()
        if p.photons:
            r.BuildPhotonMaps(p)  # GeneratePhotons is part of every BeginRender (RayTracer.cpp:24-40)
        # blocking render of this rank's sample range into `accum`, then the sample-split exchange: one RCCL
        # sum-reduce of (sum rgb, n) to rank 0 over xGMI, left in flight while the next frame renders
        pending[k] = distributed.render_partition(r, p, accum, async_op=world > 1)

    def drain():
        for q in pending:
            if q is not None:
                q.wait()
        torch.cuda.synchronize()

    if cfg.progressive_spp:
        step()   # a frame's first pass: it leaves the per-pixel lookup hints that every later pass starts from
        drain()
    # one counted, untimed step: algorithmic work per step (of a later pass, where the frame is progressive)
    r.counters(enable=True, reset=True)
    step()
    drain()
    cnt = r.counters()
    r.counters(enable=False)
    rays_per_step = cnt["rays_primary"] + cnt["rays_secondary"] + cnt["rays_shadow"]

    for _ in range(args.warmup):
        step()
    drain()
    r.profile(enable=True, reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()   # every frame's exchange has completed on rank 0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = r.profile()
    r.profile(enable=False)

    t = torch.tensor([dt], dtype=torch.float64, device=stat_dev)
    rays = torch.tensor([float(rays_per_step)], dtype=torch.float64, device=stat_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(t.item())
    total_rays_per_step = float(rays.item())

    if rank == 0:
        # final image of the last step (outside the timed region): sanity that the accumulators hold a frame
        pr = cfg.params(0)
        accum = accums[(frame[0] - 1) % len(accums)]
        img = r.resolve(pr, accum.data_ptr())
        n_expected = SPP_PER_STEP * world
        assert abs(float(img.radiance[..., 3].mean()) - n_expected) < 1e-3, "accumulator does not hold the expected samples"

        # SURVEY §8d algorithmic bytes / flops of the traversal and gather kernels: 48 B per ray or work item (32 B ray in,
        # 16 B hit / occlusion out) + 32 B per BVH box tested (two per 64-byte node fetch) + 48 B per triangle slot + 32 B
        # per photon visited; 36 flops per instance, 30 per sphere, 12 per plane, 24 per box, 51 per triangle, 20 per photon.
        # The instance table (64 B + a transform chain per instance, a few KB per scene) is NOT charged per test: every
        # lane of a wave reads the same entry through scalar loads, so its HBM traffic is the table's size, once.
        # BVH work happens in the mesh kernels (second phase); the instance table (spheres, planes, mesh world boxes) is
        # walked by the first-phase kernels.
        tc_rays = cnt["rays_primary"] + cnt["rays_secondary"]
        first_inst = cnt["closest_inst_tests"] - cnt["mesh_inst_tests"]
        sh_inst = cnt["inst_tests"] - cnt["closest_inst_tests"]          # shadow-ray instance tests, both phases
        sh_nodes = cnt["node_visits"] - cnt["closest_node_visits"]
        sh_tris = cnt["tri_tests"] - cnt["closest_tri_tests"]
        # Which kernels ran: scenes whose closest-hit rays walk the instance table (fewer than RT_TLAS_MIN = 24 instances) take
        # the FUSED first phase — k_raygen_shade / k_trace_shade trace a ray through the instance table and shade it right there
        # unless it reaches a mesh box; only those rays go through a queue, k_trace_mesh and k_shade.  Scenes with an instance
        # BVH keep k_raygen_trace / k_trace_closest and shade everything in k_shade.
        n_inst = sum(1 for i in range(scene.desc.n_nodes) if scene.desc.nodes[i].prim_kind != 0)
        fused = n_inst < int(float(os.environ.get("RT_TLAS_MIN", "24"))) and os.environ.get("RT_FUSE", "1") != "0"
        all_rays = max(tc_rays, 1)
        mesh_frac = min(1.0, cnt["mesh_items"] / all_rays) if fused else 1.0    # share of the shade points k_shade handles
        # 84 B per shade point in (k_shade only: the fused kernels shade from registers), 64 B per child ray, ~48 B per shadow
        # group (32 hard / 80 soft; groups <= shade points x lights), 16 B per texture lookup
        shade_out_bytes = 64.0 * cnt["rays_secondary"] + 48.0 * min(cnt["rays_shadow"], cnt["shades"] * scene.desc.n_lights) + 16.0 * cnt["tex_evals"]
        shade_flops = (120.0 + 40.0 * scene.desc.n_lights) * cnt["shades"] + 20.0 * cnt["tex_evals"]
        first_flops = (36.0 * first_inst + (30.0 * cnt["sphere_tests"] + 12.0 * cnt["plane_tests"]) * tc_rays
                       / max(tc_rays + cnt["rays_shadow"], 1) + (24.0 * cnt["closest_node_visits"] if cnt["mesh_items"] == 0 else 0.0))
        # The first-phase kernels are reported one by one (their launch times are separate HIP-event timers); the work the
        # counters tally for both together (instance tests, shade points, mesh hand-overs) is split by their shares of the rays
        ps = cnt["rays_primary"] / all_rays if all_rays else 1.0
        if fused:
            rest_b = 80.0 * cnt["mesh_items"] + (1.0 - mesh_frac) * shade_out_bytes
            rest_f = first_flops + (1.0 - mesh_frac) * shade_flops
            first = [("k_raygen_shade", (prof["raygen"],
                      # 12 B per path (radiance slot cleared), 80 B per ray handed to the mesh phase (64 B entry + 16 B work item),
                      # and this share of the shade points' outputs
                      12.0 * cnt["paths"] + ps * rest_b, ps * rest_f, "k_raygen_shade<false")),
                     ("k_trace_shade", (prof["trace_closest"],
                      # 64 B per secondary ray read
                      64.0 * cnt["rays_secondary"] + (1.0 - ps) * rest_b, (1.0 - ps) * rest_f, "k_trace_shade<false"))]
        else:
            tlas_b = 32.0 * cnt["closest_node_visits"] if cnt["mesh_items"] == 0 else 0.0
            first = [("k_raygen_trace", (prof["raygen"],
                      # camera rays: 80 B written per path, generated and traced in registers; the instance BVH's boxes count when it is used
                      80.0 * cnt["paths"] + ps * tlas_b, ps * first_flops, "k_raygen_trace<false")),
                     ("k_trace_closest", (prof["trace_closest"], 48.0 * cnt["rays_secondary"] + (1.0 - ps) * tlas_b,
                      (1.0 - ps) * first_flops, "k_trace_closest<false"))]
        kern = {
            first[0][0]: first[0][1],
            first[1][0]: first[1][1],
            "k_trace_mesh": (prof["trace_mesh"],
                             48.0 * cnt["mesh_items"] + 32.0 * cnt["closest_node_visits"] + 48.0 * cnt["closest_tri_tests"],
                             36.0 * cnt["mesh_inst_tests"] + 24.0 * cnt["closest_node_visits"] + 51.0 * cnt["closest_tri_tests"],
                             "k_trace_mesh<false"),
            "k_shade": (prof["shade"], 84.0 * (cnt["mesh_items"] if fused else cnt["shades"]) + mesh_frac * shade_out_bytes,
                        mesh_frac * shade_flops, "k_shade<false"),
            # shadow rays, first phase (k_shadow_hard / k_shadow_soft): every instance test is charged here
            "k_shadow": (prof["shadow"], 48.0 * cnt["rays_shadow"],
                         36.0 * sh_inst + (30.0 * cnt["sphere_tests"] + 12.0 * cnt["plane_tests"]) * cnt["rays_shadow"]
                         / max(tc_rays + cnt["rays_shadow"], 1), "k_shadow_soft<false|k_shadow_hard<false"),
            "k_shadow_mesh": (prof["shadow_mesh"], 32.0 * sh_nodes + 48.0 * sh_tris, 24.0 * sh_nodes + 51.0 * sh_tris, "k_shadow_mesh<false"),
            "k_photon_gather": (prof["photon_gather"], 32.0 * cnt["photon_visits"] + 48.0 * cnt["gathers"], 20.0 * cnt["photon_visits"],
                                "k_photon_gather"),
        }
        src_sha = kernel_source_sha()
        pmc = load_pmc_summary(cfg.name, src_sha)
        per_kernel = {}
        for name, (pk, k_bytes, k_flops, pmc_name) in kern.items():
            if not pk["launches"] or pk["ms"] <= 0:
                continue
            lps = pk["launches"] / max(args.steps, 1)
            avg_ms = pk["ms"] / pk["launches"]
            e = {"ms_per_step": pk["ms"] / max(args.steps, 1), "avg_launch_ms": avg_ms, "launches_per_step": lps,
                 "algorithmic_bytes_per_launch": k_bytes / lps, "algorithmic_flops_per_launch": k_flops / lps,
                 "hbm_algorithmic_frac": k_bytes / lps / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "valu_algorithmic_frac": k_flops / lps / (avg_ms * 1e-3) / 1e12 / VALU_PEAK_TFLOPS,
                 "hbm_traffic_bytes_per_launch": None, "hbm_traffic_frac": None, "valu_issue_util": None,
                 "active_lane_ratio": None, "occupancy_waves_per_simd": None, "l2_hit_rate": None}
            m = next((v for nm in pmc_name.split("|") for k, v in sorted((pmc or {}).items())
                      if isinstance(v, dict) and k.startswith(nm)), None)
            if m:   # counters of THIS build (the summary's kernel-source hash matches), per launch
                if m.get("hbm_bytes_per_launch") is not None:
                    e["hbm_traffic_bytes_per_launch"] = m["hbm_bytes_per_launch"]
                    e["hbm_traffic_frac"] = m["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                for k in ("valu_issue_util", "active_lane_ratio", "occupancy_waves_per_simd", "l2_hit_rate"):
                    e[k] = m.get(k)
            per_kernel[name] = e
        # BVH-node vs triangle-slot attribution of the mesh kernels.  The PMC counters see HBM traffic per
        # kernel, not per buffer; what can be stated per buffer is its algorithmic bytes (32 B per box tested, 48 B per slot
        # tested) and its resident size — the traffic measured for the whole kernel bounds what either array drew from HBM,
        # and algorithmic / measured is the reuse the caches delivered.
        n_dev_nodes = sum(max(int(scene.desc.meshes[i].n_bvh_nodes) - 1, 0) // 2 for i in range(scene.desc.n_meshes))
        resident = {"bvh_nodes": 64 * n_dev_nodes if n_dev_nodes else None, "triangle_slots": 48 * scene.n_triangles()}
        for name, nodes, tris, items in (("k_trace_mesh", cnt["closest_node_visits"], cnt["closest_tri_tests"], cnt["mesh_items"]),
                                         ("k_shadow_mesh", sh_nodes, sh_tris, 0)):
            e = per_kernel.get(name)
            if e:
                lps = e["launches_per_step"]
                e["buffers"] = {"bvh_nodes": {"algorithmic_bytes_per_launch": 32.0 * nodes / lps, "resident_bytes": resident["bvh_nodes"]},
                                "triangle_slots": {"algorithmic_bytes_per_launch": 48.0 * tris / lps, "resident_bytes": resident["triangle_slots"]},
                                "work_items": {"algorithmic_bytes_per_launch": 48.0 * items / lps}}
                if e["hbm_traffic_bytes_per_launch"]:
                    e["buffers"]["cache_reuse"] = (32.0 * nodes + 48.0 * tris) / lps / e["hbm_traffic_bytes_per_launch"]
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])   # the kernel with the most time in the timed region
        d = per_kernel[dom]
        # The binding roof: the largest MEASURED utilisation when this build's counters are at hand (HBM traffic against
        # 8 TB/s, VALU issue slots in use), otherwise the larger algorithmic fraction.  "valu": FP32 vector peak; nothing on
        # this path is a dense contraction, so there is no MFMA roof to name.
        if d["hbm_traffic_frac"] is not None and d["valu_issue_util"] is not None:
            bound = "hbm" if d["hbm_traffic_frac"] >= d["valu_issue_util"] else "valu"
        else:
            bound = "hbm" if d["hbm_algorithmic_frac"] >= d["valu_algorithmic_frac"] else "valu"
        if bound == "hbm":
            achieved, peak, unit = d["algorithmic_bytes_per_launch"] / (d["avg_launch_ms"] * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            achieved, peak, unit = d["algorithmic_flops_per_launch"] / (d["avg_launch_ms"] * 1e-3) / 1e12, VALU_PEAK_TFLOPS, "TFLOP/s"
        kernel_ms = {k: round(v["ms"] / max(args.steps, 1), 3) for k, v in prof.items() if v["launches"]}
        out = {
            "metric": "Mrays/s", "value": total_rays_per_step * args.steps / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.workload}; {WIDTH}x{HEIGHT}, {SPP_PER_STEP} spp per GPU per step, Philox seed 1",
                       "triangles": scene.n_triangles(), "mesh_bvh": "binned SAH, host loader" if HOST_BVH else "linear BVH, built on the device at commit",
                       "scene_load_s": round(scene_load_s, 3), "scene_commit_s": round(scene_commit_s, 3),
                       "rays_per_step_per_gpu": rays_per_step, "paths_per_step_per_gpu": cnt["paths"],
                       "mpaths_per_s": cnt["paths"] * world * args.steps / dt / 1e6,
                       # soft-shadow groups whose outcome is proven are finished without their rays: `value` counts
                       # only rays that were traced; with the rays the reference traces for the same frame it would be
                       "rays_proven_per_step_per_gpu": cnt["rays_shadow_proven"],
                       "reference_equivalent_mrays_s": (rays_per_step + cnt["rays_shadow_proven"]) * world * args.steps / dt / 1e6,
                       "partition": "sample-range split; RCCL sum-reduce of (sum rgb, n) per frame, overlapped with the next frame's rendering" if world > 1 else "single GPU",
                       # what the exchange actually ran over (N > 1): the process group's backend and the ranks it saw
                       "collective_backend": (dist.get_backend() if world > 1 else None), "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                       "first_phase_fused_with_shading": fused,
                       "kernel_ms_per_step": kernel_ms, "counters_per_step": cnt},
            "roofline": {"bound": bound, "kernel": dom, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": d["hbm_traffic_bytes_per_launch"],
                         # what THIS build's counters say about the same roof (null without a matching PMC summary): HBM traffic
                         # against 8 TB/s, or the share of VALU issue slots in use (at active_lane_ratio lanes per instruction)
                         "measured_util": d["hbm_traffic_frac"] if bound == "hbm" else d["valu_issue_util"],
                         "active_lane_ratio": d["active_lane_ratio"],
                         "avg_launch_ms": d["avg_launch_ms"], "launches_per_step": d["launches_per_step"],
                         "kernel_source_sha": src_sha, "pmc_summary": (pmc or {}).get("_meta", {}).get("file"),
                         "per_kernel": per_kernel,
                         "note": "achieved = SURVEY 8d algorithmic bytes (bound hbm) or flops (bound valu) per launch / HIP-event launch time; "
                                 "traffic = (2 x FETCH_SIZE + WRITE_SIZE) per launch from this build's PMC passes (null when no summary under "
                                 "profiles/ carries this kernel-source hash); per_kernel lists, for every hot kernel, the algorithmic and the "
                                 "measured HBM fraction, the algorithmic VALU fraction, VALU issue-slot utilisation, active lanes and occupancy"},
        }
        if brief:
            row = {"workload": cfg.workload, "frame": f"{WIDTH}x{HEIGHT}, {SPP_PER_STEP} spp per step", "ms_per_step": out["ms_per_step"],
                   "mrays_per_s": out["value"], "mpaths_per_s": out["config"]["mpaths_per_s"],
                   "reference_equivalent_mrays_s": out["config"]["reference_equivalent_mrays_s"],
                   "triangles": scene.n_triangles(), "scene_load_s": round(scene_load_s, 3), "scene_commit_s": round(scene_commit_s, 3),
                   "dominant_kernel": dom, "dominant_kernel_ms_per_step": d["ms_per_step"], "bound": bound,
                   "roofline_frac_algorithmic": achieved / peak,
                   "measured_util": d["hbm_traffic_frac"] if bound == "hbm" else d["valu_issue_util"],
                   "active_lane_ratio": d["active_lane_ratio"], "pmc_summary": (pmc or {}).get("_meta", {}).get("file"),
                   "kernel_ms_per_step": kernel_ms,
                   "vs_oracle": psnr_vs_oracle(cfg, scene, r, 4 * cfg.cpu_tile_stride if cfg.cpu_tile_stride > 1 else 1)}
            r.close()
            return row
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, scene, r)
        r.close()
        return out
    r.close()
    return None


if __name__ == "__main__":
    main()
