"""GPU parity of the BASELINE configurations the round-1 suite did not run on hardware, through the C-ABI against the CPU
oracle: the Monte-Carlo indirect path (C3), adaptive sampling (C4's single-GPU mode), depth of field / environment /
nested textures, the stress-scene generator (C5, scaled and full size) and the asynchronous Begin / poll / Stop API."""
import os

import numpy as np
import pytest

import common as cc
from common import params
from conftest import ROOT, scene_path
from raytracer_amd import RT_HIT_FRONT, Scene

pytestmark = pytest.mark.gpu


def test_c3_monte_carlo_indirect(gpu_renderer):
    """BASELINE config 3: textured Cornell box, SampleIndirectDiffuseCosin x 2 bounces + 3 specular bounces + soft shadows
    (materials.cpp:175-212, :309-314), radiance compared at the same Philox seed."""
    s = cc.check_monte_carlo_indirect(gpu_renderer, Scene.load(os.path.join(ROOT, "scenes", "c3_cornell_textured.xml")), 320, 180, spp=4)
    print("c3 mc:", s)


def test_monte_carlo_on_textured_scene(gpu_renderer):
    s = cc.check_monte_carlo_indirect(gpu_renderer, Scene.from_string(cc.TEXTURED), 192, 128, spp=4)
    print("textured mc:", s)


def test_textures_dof_environment_direct_light(gpu_renderer):
    s = cc.check_textured_scene(gpu_renderer, 384, 256, spp=4)
    print("textured:", s)


def test_adaptive_sampling_4_24(gpu_renderer):
    f = cc.check_adaptive_sampling(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), 240, 135, 4, 24, 0.02, equal_frac=0.9995)
    print("adaptive 4..24: equal sample counts for", f)


def test_adaptive_sampling_reference_32_128(gpu_renderer):
    """The reference's own constants: minSamples 32, maxSamples 128, threshold 0.01 (raytracer.h:20-21, RayTracer.cpp:191)."""
    f = cc.check_adaptive_sampling(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), 96, 54, 32, 128, 0.01, equal_frac=0.9995)
    print("adaptive 32..128: equal sample counts for", f)


def test_async_render_equals_blocking_render(gpu_renderer):
    seen = cc.check_async_render(gpu_renderer, Scene.load(scene_path("c2_mesh_room.xml")), 960, 540, spp=8)
    assert len(set(seen)) >= 3, f"rt_render_poll never showed a partial frame: {sorted(set(seen))}"


def test_c5_stress_scene_scaled(gpu_renderer):
    """BASELINE config 5's generator at 1/55 of its size (227k triangles in one BVH): ids bit-exact, batched parity, and a
    stochastic frame with the two sphere lights' soft shadows."""
    import oracle_ffi
    scene = Scene.stress(seed=7, n_instances=96, subdiv=3, ground_res=230, width=480, height=270)
    assert 200000 < scene.n_triangles() < 300000
    cc.check_stress_scene(gpu_renderer, scene, 480, 270, n_rays=20000)
    p = params(240, 135, spp=2, jitter=1, bounces=3)
    g, c = gpu_renderer.Render(p), oracle_ffi.OracleScene(scene).render(p)
    print("c5 scaled:", cc.assert_stochastic_close(g, c, "stress scene"))


def test_shadow_hints_change_the_walk_only(gpu_renderer):
    """k_shadow_mesh with and without occluder hints on the scaled stress scene (host SAH tree and device-built tree)."""
    scene = Scene.stress(seed=7, n_instances=96, subdiv=3, ground_res=230, width=480, height=270)
    off, on = cc.check_shadow_hints(gpu_renderer, scene, 240, 135, spp=2)
    assert on < off
    bare = Scene.stress(seed=7, n_instances=96, subdiv=3, ground_res=230, width=480, height=270, host_bvh=False)
    cc.check_shadow_hints(gpu_renderer, bare, 160, 90, device_bvh=True)
    cc.check_shadow_hints(gpu_renderer, cc.load_multimtl_scene(), 160, 90, spp=2, tol=1e-4)   # six mesh instances, one sized light


def test_soft_shadow_clear_view_shortcut(gpu_renderer):
    """Soft-shadow groups with a proven clear view of the light are fully lit, and groups in the full shadow of a sphere dark,
    without tracing (cone_clear): same frame as with every group traced, oracle parity, and fewer rays — on the Cornell box,
    C3's scene, random nested scene graphs and the scaled stress scene (one big mesh: no group can be cleared, nothing may
    change)."""
    for name, w, h in (("cornell_box.xml", 320, 240), ("c3_cornell_textured.xml", 320, 180)):
        n1, n0 = cc.check_soft_shadow_clear_view(gpu_renderer, Scene.load(scene_path(name)), w, h, spp=2, tol=1e-5)
        print(name, "shadow rays", n1, "of", n0)
    print("random scene graphs: shadow rays %d of %d" % cc.check_soft_shadow_clear_view_random(gpu_renderer, range(0, 12), 160, 120, tol=1e-5))
    cb = Scene.load(scene_path("cornell_box.xml"))   # a wider light, other sample counts (min = max: one phase only)
    cc.check_soft_shadow_clear_view(gpu_renderer, cb, 320, 240, tol=1e-5, light_scale=3.0)
    cc.check_soft_shadow_clear_view(gpu_renderer, cb, 320, 240, tol=1e-5, shadow=(4, 40))
    cc.check_soft_shadow_clear_view(gpu_renderer, cb, 320, 240, tol=1e-5, shadow=(8, 8))
    stress = Scene.stress(seed=7, n_instances=24, subdiv=3, ground_res=96, width=160, height=90)
    cc.check_soft_shadow_clear_view(gpu_renderer, stress, 160, 90, tol=1e-5, min_saved=-1.0)


def test_soft_shadow_proofs_through_the_instance_bvh(gpu_renderer):
    """finalProject.xml (the scene the reference's main.cpp hard-codes; 70 instances) and a Cornell box forced both ways."""
    print("proofs, many instances:", cc.check_soft_shadow_proofs_many_instances(gpu_renderer, 320, 180, tol=1e-5))


def test_soft_shadow_proofs_full_size(gpu_renderer):
    """BASELINE configs 3 and 4 at their full frame sizes with their own integrator settings (C3: 4K, Monte-Carlo indirect, three
    bounces; C4's Cornell box at 1080p): the frame with the proofs equals the frame that traces every group — ids exactly,
    radiance to 1e-5 — and traced + proven rays add up to the rays of the traced frame."""
    n1, n0 = cc.check_soft_shadow_clear_view(gpu_renderer, Scene.load(scene_path("c3_cornell_textured.xml")), 3840, 2160, tol=1e-5,
                                             with_oracle=False, bounces=3, mc=2)
    print("C3 4K x 1: shadow rays", n1, "of", n0)
    assert n1 < 0.25 * n0
    n1, n0 = cc.check_soft_shadow_clear_view(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), 1920, 1080, spp=2, tol=1e-5,
                                             with_oracle=False, bounces=3)
    print("Cornell box 1080p x 2: shadow rays", n1, "of", n0)


def test_device_bvh_build(gpu_renderer):
    """n2: meshes without a tree get a linear BVH on the device at rt_scene_commit (rt_bvh_build.h): a valid BVH of the mesh,
    and with the oracle walking the SAME tree (rt_scene_mesh_bvh) node + face ids are bit-exact and the batched closest /
    any-hit queries agree — on the scaled stress scene (227k triangles in one mesh) and on C2's instanced meshes."""
    import oracle_ffi
    scene = Scene.stress(seed=7, n_instances=96, subdiv=3, ground_res=230, width=480, height=270)
    trees, depths = cc.check_device_bvh(gpu_renderer, scene, 480, 270, n_rays=20000)
    print("device BVH of the scaled stress scene: depth", depths, "nodes", [len(t[0]) for t in trees.values()])
    c2 = Scene.load(scene_path("c2_mesh_room.xml"))
    bare = c2.with_bvh({i: None for i in range(c2.desc.n_meshes)})
    gpu_renderer.LoadScene(bare)
    built = {i: gpu_renderer.GetMeshBVH(i) for i in range(c2.desc.n_meshes)}
    for i in built:
        cc.assert_bvh_valid(built[i][0], built[i][1], c2.desc.meshes[i])
    p = params(480, 270, spp=1, jitter=0, bounces=3)
    g, c = gpu_renderer.Render(p), oracle_ffi.OracleScene(c2.with_bvh(built)).render(p)
    assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face)
    assert cc.rel_colour_diff(g.radiance, c.radiance) <= 1e-4


def test_c5_stress_scene_full_size(gpu_renderer):
    """BASELINE config 5 at full size (12.6 M triangles in one BVH, 4K): properties that need no full CPU render — no
    queue overflow / walk abort (the call succeeds), every pixel sampled, the frame's two-phase kernels and the batched
    one-phase walk agree on node AND face ids for the same camera rays, and the oracle agrees on a tile subset."""
    import oracle_ffi
    scene = Scene.stress(seed=7, n_instances=2048, subdiv=4, ground_res=1024, width=3840, height=2160)
    assert scene.n_triangles() > 12_000_000
    gpu_renderer.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    p = params(3840, 2160, spp=1, jitter=0, bounces=0, light_scale=0.0)   # (one shadow ray per light: the oracle walks 12.6 M triangles too)
    g = gpu_renderer.Render(p)
    assert (g.sample_count == 1).all()
    assert (g.hit_face >= 0).mean() > 0.2
    # the same camera rays through the batched entry point (pixel centres of every 37th row, every 11th column)
    ys, xs = np.arange(5, 2160, 37), np.arange(3, 3840, 11)
    rays = np.zeros(len(ys) * len(xs), dtype=cc.rays_from_arrays(np.zeros((1, 3)), np.zeros((1, 3))).dtype)
    k = 0
    for y in ys[:: max(1, len(ys) // 12)]:
        for x in xs:
            o, d = orc.primary_ray(p, int(x), int(y), 0)
            rays["p"][k], rays["dir"][k] = o, d
            k += 1
    rays = rays[:k]
    yy = np.repeat(ys[:: max(1, len(ys) // 12)], len(xs))[:k]
    xx = np.tile(xs, len(ys))[:k]
    hits = gpu_renderer.TraceRay(rays, RT_HIT_FRONT)
    assert np.array_equal(hits["node"], g.hit_node[yy, xx]) and np.array_equal(hits["face"], g.hit_face[yy, xx])
    # the oracle on every 509th 32x32 tile (16 tiles): ids bit-exact, colour within 1e-4
    q = params(3840, 2160, spp=1, jitter=0, bounces=0, light_scale=0.0)
    q.tile_begin, q.tile_stride = 0, 509
    c = orc.render(q)
    covered = c.sample_count > 0
    assert covered.sum() >= 16 * 1024 * 0.9
    assert np.array_equal(g.hit_node[covered], c.hit_node[covered]) and np.array_equal(g.hit_face[covered], c.hit_face[covered])
    assert cc.rel_colour_diff(g.radiance[covered], c.radiance[covered]) <= 1e-4
    # shadow rays: batched any-hit against the oracle on the same rays
    tmax = np.full(len(rays), np.finfo(np.float32).max, np.float32)
    sub = slice(0, 1500)
    assert np.array_equal(gpu_renderer.TraceShadowRay(rays[sub], tmax[sub]), orc.trace_any(rays[sub], tmax[sub]))


def test_steady_state_calls_do_not_allocate(gpu_renderer):
    """no hipMalloc / hipFree in a steady-state rt_render / rt_render_accumulate / rt_resolve /
    rt_trace_* / rt_denoise (the buffers live in the context, grow-only)."""
    import torch
    scene = Scene.load(scene_path("c2_mesh_room.xml"))
    gpu_renderer.LoadScene(scene)
    p = params(480, 270, spp=4, jitter=1, bounces=3)
    rays = cc.random_rays(5000, seed=3)
    tmax = np.full(len(rays), 1.0, np.float32)
    acc = torch.zeros(480 * 270 * 8, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()

    def calls():
        gpu_renderer.Render(p)
        gpu_renderer.render_accumulate(p, acc.data_ptr())
        gpu_renderer.resolve(p, acc.data_ptr())
        gpu_renderer.TraceRay(rays)
        gpu_renderer.TraceShadowRay(rays, tmax)
        gpu_renderer.Denoise(np.zeros((270, 480, 3), np.float32))

    calls()
    before = gpu_renderer.alloc_count()
    for _ in range(3):
        calls()
    assert gpu_renderer.alloc_count() == before


def test_reference_scene_fixtures(gpu_renderer):
    """Sixteen bundled reference scenes (scene data fixtures; scene3_2 = scene3 and cornellBox = cornell_box.xml make
    eighteen) on the GPU against the oracle: ids bit-exact, colour 1e-4,
    stochastic frames within the gate of assert_stochastic_close."""
    cc.check_reference_scene_fixtures(gpu_renderer, 320, 180)


def test_multimtl_scene(gpu_renderer):
    """n1: OBJ + MTL pairs on nodes without material= (CreateMultiMtl, xmlload.cpp:319-343; sub-material 0) on the
    GPU against the oracle."""
    print("multimtl:", cc.check_multimtl_scene(gpu_renderer, 480, 270))


def test_reduce_accumulators_over_rccl_single_rank(gpu_renderer):
    """rt_reduce_accumulators (the C hosts' multi-GPU exchange): a one-rank RCCL communicator made through ctypes, the
    caller's own ncclReduce handed over as a function pointer; a one-rank sum-reduce leaves the accumulator as it is.
    (The N > 1 exchange itself is the driver's scaling run; the partition logic is covered by the gloo tests.)"""
    import ctypes as C
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        path = "/opt/rocm/lib/librccl.so"
    rccl = C.CDLL(path)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        w, h = 64, 32
        acc = torch.arange(w * h * 8, dtype=torch.float32, device="cuda:0")
        want = acc.clone()
        torch.cuda.synchronize()
        fn = C.cast(rccl.ncclReduce, C.c_void_p).value
        gpu_renderer.reduce_accumulators(comm.value, acc.data_ptr(), w, h, root=0, nccl_reduce_fn=fn)
        assert torch.equal(acc, want)
        with pytest.raises(RuntimeError):
            gpu_renderer.reduce_accumulators(0, acc.data_ptr(), w, h)
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
