"""CPU-side logic check of the integrator: the HOST EMULATION build of raytracer_amd/csrc (tests/_emu, g++ -DRT_EMU,
single-threaded) against the recursive CPU oracle.  Exercises the queue-per-stage unrolling of the reference's
recursion, the shadow-group phases, adaptive sampling, partitioning and the batched trace API where no GPU exists;
the same comparisons run against the real HIP library in test_gpu_*.py."""
import ctypes as C

import numpy as np
import pytest

import emu_ffi
import oracle_ffi
import os

import common as common_checks
from common import TEXTURED, assert_hits_equal, params, psnr8, random_rays, rel_colour_diff
from conftest import ROOT, scene_path
from raytracer_amd import RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK, Scene

@pytest.fixture(scope="module")
def emu():
    return emu_ffi.EmuRenderer()


@pytest.mark.parametrize("name", ["spheres_room.xml", "cornell_box.xml", "c2_mesh_room.xml"])
def test_trace_api(emu, name):
    scene = Scene.load(scene_path(name))
    emu.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    rays = random_rays(3000, seed=21)
    for side in (RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK):
        assert_hits_equal(emu.TraceRay(rays, side), orc.trace_closest(rays, side), name)
    tmax = (0.2 + 2.0 * np.random.default_rng(2).random(len(rays))).astype(np.float32)
    assert np.array_equal(emu.TraceShadowRay(rays, tmax), orc.trace_any(rays, tmax))


@pytest.mark.parametrize("name", ["spheres_room.xml", "cornell_box.xml", "c2_mesh_room.xml"])
def test_config1_frames_bit_exact_ids(emu, name):
    scene = Scene.load(scene_path(name))
    emu.LoadScene(scene)
    p = params(160, 120, spp=1, jitter=0, bounces=0, light_scale=0.0)
    g, c = emu.Render(p), oracle_ffi.OracleScene(scene).render(p)
    assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face)
    assert rel_colour_diff(g.radiance, c.radiance) <= 1e-6


@pytest.mark.parametrize("name,spp,bounces", [("cornell_box.xml", 3, 3), ("c2_mesh_room.xml", 2, 3)])
def test_recursion_unrolled_into_queues(emu, name, spp, bounces):
    scene = Scene.load(scene_path(name))
    emu.LoadScene(scene)
    p = params(96, 54, spp=spp, jitter=1, bounces=bounces)
    g, c = emu.Render(p), oracle_ffi.OracleScene(scene).render(p)
    assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= 2e-5
    assert psnr8(g.img, c.img) >= 50


def test_textures_dof_environment_direct_light_white_material(emu):
    common_checks.check_textured_scene(emu, 96, 64, spp=3, strict=5e-5)


def test_monte_carlo_indirect(emu):
    common_checks.check_monte_carlo_indirect(emu, Scene.from_string(TEXTURED), 64, 48, spp=2, strict=5e-5)


def test_monte_carlo_indirect_config3_scene(emu):
    """BASELINE config 3's scene and integrator (checker floor, brick wall, cloud environment, mc = 2, soft shadows)."""
    common_checks.check_monte_carlo_indirect(emu, Scene.load(os.path.join(ROOT, "scenes", "c3_cornell_textured.xml")), 40, 24, spp=2,
                                             strict=5e-5)


def test_adaptive_sampling_matches_sequential_loop(emu):
    """RayTracer.cpp:136-204: the stop test after every sample >= spp_min; Q1/Q2 positions."""
    common_checks.check_adaptive_sampling(emu, Scene.load(scene_path("cornell_box.xml")), 48, 27, 4, 24, 0.02, strict=2e-5)


def test_async_render_equals_blocking_render(emu):
    seen = common_checks.check_async_render(emu, Scene.load(scene_path("cornell_box.xml")), 160, 120, spp=2)
    assert len(set(seen)) >= 3, f"rt_render_poll never showed a partial frame: {sorted(set(seen))}"


def test_stress_scene_scaled(emu):
    """BASELINE config 5's generator at ~1/250 of its size."""
    scene = Scene.stress(seed=7, n_instances=24, subdiv=3, ground_res=96, width=160, height=90)
    assert 45000 < scene.n_triangles() < 60000
    common_checks.check_stress_scene(emu, scene, 160, 90, n_rays=1500)
    # soft shadows of the two sphere lights through the mesh phase: 16 samples, then the other 112 in chunks (run_pass)
    p = params(48, 27, spp=1, jitter=1, bounces=1)
    g, c = emu.Render(p), oracle_ffi.OracleScene(scene).render(p)
    assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= 5e-5


def test_shadow_hints_change_the_walk_only(emu):
    scene = Scene.stress(seed=7, n_instances=24, subdiv=3, ground_res=96, width=160, height=90)
    off, on = common_checks.check_shadow_hints(emu, scene, 64, 36)
    assert on < off   # fewer node visits to the same answers
    bare = Scene.stress(seed=7, n_instances=24, subdiv=3, ground_res=96, width=160, height=90, host_bvh=False)
    common_checks.check_shadow_hints(emu, bare, 48, 27, device_bvh=True)   # the device-built tree names its slot ranges too
    # several mesh instances under one sized light: a hint only steers the walk of the instance it was found in
    common_checks.check_shadow_hints(emu, common_checks.load_multimtl_scene(), 80, 45, tol=1e-4)


def test_soft_shadow_clear_view_shortcut(emu):
    """Cornell box with a sphere light (nested transforms, planes and spheres) and the textured C3 scene."""
    n1, n0 = common_checks.check_soft_shadow_clear_view(emu, Scene.load(scene_path("cornell_box.xml")), 64, 48, strict=1e-4)
    print("cornell_box: shadow rays", n1, "of", n0)
    common_checks.check_soft_shadow_clear_view(emu, Scene.load(scene_path("c3_cornell_textured.xml")), 48, 27, strict=1e-4)
    print("random scene graphs: shadow rays %d of %d" % common_checks.check_soft_shadow_clear_view_random(emu, range(4, 10), 64, 48, strict=1e-4))
    # a light three times the size (wider cones), and sample counts other than 16 / 128 — among them min = max, where every
    # group ends after the first phase
    cb = Scene.load(scene_path("cornell_box.xml"))
    common_checks.check_soft_shadow_clear_view(emu, cb, 64, 48, strict=1e-4, light_scale=3.0)
    common_checks.check_soft_shadow_clear_view(emu, cb, 64, 48, strict=1e-4, shadow=(4, 40))
    common_checks.check_soft_shadow_clear_view(emu, cb, 64, 48, strict=1e-4, shadow=(8, 8))


def test_soft_shadow_proofs_through_the_instance_bvh(emu):
    print(common_checks.check_soft_shadow_proofs_many_instances(emu, 64, 36, strict=1e-4))


def test_device_bvh_build(emu, tmp_path):
    """Meshes without a tree get a linear BVH at rt_scene_commit (here: the same kernels under host emulation); a tree that
    comes with the mesh is handed back unchanged by rt_scene_mesh_bvh."""
    scene = Scene.stress(seed=3, n_instances=12, subdiv=2, ground_res=40, width=96, height=54)
    trees, depths = common_checks.check_device_bvh(emu, scene, 96, 54, n_rays=800)
    assert 8 <= depths[0] <= 28
    # Scene.stress(host_bvh=False) hands the mesh over without a tree in the first place
    bare = Scene.stress(seed=3, n_instances=12, subdiv=2, ground_res=40, width=96, height=54, host_bvh=False)
    assert not bare.desc.meshes[0].bvh and bare.desc.meshes[0].n_bvh_nodes == 0
    emu.LoadScene(bare)
    again = emu.GetMeshBVH(0)
    assert np.array_equal(again[0].view(np.uint32), trees[0][0].view(np.uint32)) and np.array_equal(again[1], trees[0][1])
    # a scene with both kinds of mesh: the first keeps the tree it came with, the second gets one built on the device
    c2 = Scene.load(scene_path("c2_mesh_room.xml"))
    assert c2.desc.n_meshes == 2
    mixed = c2.with_bvh({1: None})
    emu.LoadScene(mixed)
    kept, built = emu.GetMeshBVH(0), emu.GetMeshBVH(1)
    m0 = c2.desc.meshes[0]
    assert np.array_equal(kept[0].view(np.uint32), np.ctypeslib.as_array(C.cast(m0.bvh, C.POINTER(C.c_uint32)), shape=(m0.n_bvh_nodes * 8,)))
    common_checks.assert_bvh_valid(built[0], built[1], c2.desc.meshes[1])
    p = params(64, 36, spp=1, jitter=0, bounces=2)
    g, c = emu.Render(p), oracle_ffi.OracleScene(c2.with_bvh({1: built})).render(p)
    assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face)
    assert common_checks.rel_colour_diff(g.radiance, c.radiance) <= 1e-4
    # tiny meshes: one leaf, and a mesh of exactly 5 faces (one split); many faces with the same centre (equal codes)
    for n_faces, same in ((1, False), (4, False), (5, False), (40, True)):
        with open(tmp_path / f"tiny{n_faces}.obj", "w") as f:
            for k in range(n_faces):
                x = 0.0 if same else 1.5 * k
                s_ = 1.0 + (0.01 * k if same else 0.0)
                f.write(f"v {x - s_} {-s_} 0\nv {x + s_} {-s_} {0.1 * k}\nv {x + s_} {s_} 0\nv {x - s_} {s_} {0.1 * k}\n")
            for k in range(n_faces):
                f.write(f"f {4 * k + 1} {4 * k + 2} {4 * k + 4}\n" if not same else f"f {4 * k + 1} {4 * k + 3} {4 * k + 2}\n")
        tiny = Scene.from_string(f"""<xml><scene><object type="obj" name="tiny{n_faces}.obj" material="m"/>
<material type="blinn" name="m"><diffuse value="0.5"/></material><light type="direct"><intensity value="1"/><direction x="0" y="0" z="-1"/></light>
</scene><camera><position x="0" y="-1" z="30"/><target x="0" y="0" z="0"/><up x="0" y="1" z="0"/><fov value="40"/>
<width value="32"/><height value="32"/></camera></xml>""", asset_dir=str(tmp_path), host_bvh=False)
        assert tiny.desc.meshes[0].n_faces == n_faces
        emu.LoadScene(tiny)
        nodes, elems = emu.GetMeshBVH(0)
        common_checks.assert_bvh_valid(nodes, elems, tiny.desc.meshes[0])
        assert len(nodes) == (1 if n_faces <= 4 else 3) or same
        p = params(32, 32, spp=1, jitter=0, bounces=0)
        g, c = emu.Render(p), oracle_ffi.OracleScene(tiny.with_bvh({0: (nodes, elems)})).render(p)
        assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face) and (g.hit_face >= 0).any()


def test_steady_state_calls_do_not_allocate(emu):
    """Entry points keep their device buffers in the context (grow-only): repeating a call allocates nothing."""
    scene = Scene.load(scene_path("cornell_box.xml"))
    emu.LoadScene(scene)
    p = params(96, 54, spp=2, jitter=1, bounces=2)
    rays = random_rays(500, seed=3)
    tmax = np.full(len(rays), 1.0, np.float32)
    acc = np.zeros(96 * 54 * 8, np.float32)

    def calls():
        emu.Render(p)
        emu.render_accumulate(p, acc.ctypes.data)
        emu.resolve(p, acc.ctypes.data)
        emu.TraceRay(rays)
        emu.TraceShadowRay(rays, tmax)
        emu.Denoise(np.zeros((54, 96, 3), np.float32))

    calls()
    before = emu.alloc_count()
    assert before > 0
    for _ in range(3):
        calls()
    assert emu.alloc_count() == before


def test_tile_and_sample_partitions_reproduce_the_frame(emu):
    """partition logic with G 'virtual devices' — tile split is exact, sample split sums to the frame."""
    scene = Scene.load(scene_path("cornell_box.xml"))
    emu.LoadScene(scene)
    W, H = 80, 72
    p = params(W, H, spp=4, jitter=1, bounces=1)
    full = emu.Render(p)
    G = 3
    acc = np.zeros(W * H * 8, np.float32)
    for r in range(G):
        q = params(W, H, spp=4, jitter=1, bounces=1)
        q.tile_begin, q.tile_stride = r, G
        emu.render_accumulate(q, acc.ctypes.data)
    tiles = emu.resolve(p, acc.ctypes.data)
    assert np.array_equal(tiles.radiance, full.radiance) and np.array_equal(tiles.img, full.img)
    acc[:] = 0
    for r in range(2):
        q = params(W, H, spp=4, jitter=1, bounces=1)
        q.sample_begin, q.sample_count = 2 * r, 2
        emu.render_accumulate(q, acc.ctypes.data)
    split = emu.resolve(p, acc.ctypes.data)
    assert np.allclose(split.radiance, full.radiance, rtol=1e-6, atol=1e-7)
    # and the oracle agrees on what a sample range is
    q = params(W, H, spp=4, jitter=1, bounces=1)
    q.sample_begin, q.sample_count = 2, 2
    o = oracle_ffi.OracleScene(scene).render(q)
    acc[:] = 0
    emu.render_accumulate(q, acc.ctypes.data)
    assert np.abs(emu.resolve(q, acc.ctypes.data).radiance - o.radiance).max() <= 2e-5


def test_counters_agree_with_oracle_on_ray_counts(emu):
    scene = Scene.load(scene_path("cornell_box.xml"))
    emu.LoadScene(scene)
    p = params(64, 36, spp=2, jitter=1, bounces=3)
    p.skip_zero_weight = 1
    emu.counters(enable=True, reset=True)
    emu.Render(p)
    g = emu.counters()
    emu.counters(enable=False)
    _, c = oracle_ffi.OracleScene(scene).render(p, want_counters=True)
    for k in ("rays_primary", "rays_secondary", "shades", "paths"):
        assert g[k] == c[k], k
    assert g["rays_shadow"] <= c["rays_shadow"]  # the device skips shadow rays whose unshadowed term is exactly zero


def test_error_paths(emu):
    scene = Scene.load(scene_path("spheres_room.xml"))
    emu.LoadScene(scene)
    p = params(16, 16)
    p.sampler = 1
    with pytest.raises(RuntimeError, match="Philox"):
        emu.Render(p)
    p = params(16, 16)
    p.max_bounces = 9
    with pytest.raises(RuntimeError):
        emu.Render(p)
    fresh = emu_ffi.EmuRenderer()
    with pytest.raises(RuntimeError, match="no scene"):
        fresh.TraceRay(random_rays(4, 1))


def test_undersized_queues_split_the_pass_and_still_match(monkeypatch):
    """Queue capacities come from a measured sizing pass; a pass that overflows anyway leaves the accumulator alone and
    is re-run in halves.  Forced here by shrinking the measured fan-out 20x: the frame must still equal the oracle's."""
    r = emu_ffi.EmuRenderer()
    r.set_option("fan_scale", 0.05)
    scene = Scene.load(scene_path("cornell_box.xml"))
    r.LoadScene(scene)
    p = params(128, 128, spp=4, jitter=1, bounces=2)
    p.frame_mem_mb = 1  # the smallest pass size (65536 camera samples): this frame is exactly one full pass
    r.counters(enable=True, reset=True)
    r.profile(enable=True, reset=True)
    g = r.Render(p)
    cnt = r.counters()
    assert r.profile()["raygen"]["launches"] > 3, "the pass was not split: the test did not reach the overflow path"
    c, ccnt = oracle_ffi.OracleScene(scene).render(p, want_counters=True)
    assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= 2e-5
    assert (g.radiance[..., 3] == 4).all()
    for k in ("rays_primary", "rays_secondary", "paths", "shades"):  # work of overflowed attempts is not counted
        assert cnt[k] == ccnt[k], k


def test_geometry_buffers_match_oracle(emu):
    """rt_framebuffer depth / normal / albedo of the first sample's primary hit (RenderImage::zbuffer, renderer.h:39; the
    guide images of the Denoise(color, albedo, normal) overload, denoiser.h:16-17)."""
    common_checks.check_geometry_buffers(emu, Scene.load(scene_path("c2_mesh_room.xml")), 96, 54)
    common_checks.check_geometry_buffers(emu, Scene.load(os.path.join(ROOT, "scenes", "c3_cornell_textured.xml")), 96, 54)


def test_instance_bvh_matches_scene_graph_walk(emu, monkeypatch):
    common_checks.check_instance_bvh(emu, monkeypatch, scene_path)


def test_extended_materials(emu):
    common_checks.check_extended_materials(emu, w=96, h=54, spp=2)


def test_texture_filter(emu):
    common_checks.check_texture_filter(emu, w=96, h=54, ref_spp=32)


def test_boundary_extras(emu):
    common_checks.check_boundary_extras(emu)


def test_edge_cases(emu):
    common_checks.check_edge_cases(emu)


def test_random_scene_graphs(emu):
    common_checks.check_random_scene_graphs(emu, seeds=(1, 2, 3), n_rays=1500, w=64, h=48)


REF_SCENES = "/root/reference/RayTracer/scenes"


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference checkout not present on this machine")
def test_every_bundled_reference_scene_renders_like_the_oracle(emu):
    """north_star: every bundled scene renders within tolerance of the CPU reference.  The reference's own scene files
    (read in place; meshes whose .obj is absent from the checkout are skipped with a warning) through the
    emulation build of the kernels against the oracle: primary hit ids bit-exact and colour within 1e-4 at the pixel
    centre, PSNR >= 40 dB with jitter, recursion, soft shadows and textures on."""
    import glob
    files = sorted(glob.glob(os.path.join(REF_SCENES, "*.xml"))) + [os.path.join(REF_SCENES, "minecraft_world", "mc_world.xml")]
    assert len(files) >= 17
    for f in files:
        scene = Scene.load(f)
        emu.LoadScene(scene)
        orc = oracle_ffi.OracleScene(scene)
        p = params(48, 27, spp=1, jitter=0, bounces=0)
        g, c = emu.Render(p), orc.render(p)
        assert np.array_equal(g.hit_node, c.hit_node), f
        assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4, f
        p = params(32, 18, spp=2, jitter=1, bounces=2)
        p.shadow_min, p.shadow_max = 4, 8
        g, c = emu.Render(p), orc.render(p)
        assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= 1e-4, f
        assert psnr8(g.img, c.img) >= 60.0, f"{f}: PSNR {psnr8(g.img, c.img):.1f}"


def test_philox_known_answers_through_the_kernel_source(emu):
    common_checks.check_philox_on_device(emu, n=5000)


def test_multimtl_scene(emu):
    common_checks.check_multimtl_scene(emu, 96, 54, strict=1e-4)


def test_reference_scene_fixtures(emu):
    common_checks.check_reference_scene_fixtures(emu, 64, 36, strict=1e-4)
