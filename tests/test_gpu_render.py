"""GPU parity: whole frames through rt_render against the CPU oracle."""
import numpy as np
import pytest

import oracle_ffi
import os

from common import assert_stochastic_close, check_boundary_extras, check_extended_materials, check_texture_filter, check_edge_cases, check_philox_on_device, check_random_scene_graphs, check_geometry_buffers, check_instance_bvh, params, psnr8, rel_colour_diff
from conftest import ROOT, scene_path
from raytracer_amd import Scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["spheres_room.xml", "cornell_box.xml"])
def test_config1_deterministic_800x600(gpu_renderer, name):
    """BASELINE config 1: spheres+planes, Blinn + point-light shadows, 1 spp at the pixel centre, 800x600.
    Gate: bit-exact primary hit ids, colour within 1e-4 relative of the CPU oracle."""
    scene = Scene.load(scene_path(name))
    gpu_renderer.LoadScene(scene)
    p = params(800, 600, spp=1, jitter=0, bounces=0, light_scale=0.0)
    g = gpu_renderer.Render(p)
    c = oracle_ffi.OracleScene(scene).render(p)
    assert np.array_equal(g.hit_node, c.hit_node)
    assert np.array_equal(g.sample_count, c.sample_count)
    assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4
    assert np.abs(g.img.astype(int) - c.img.astype(int)).max() <= 1
    assert (c.hit_node >= 0).mean() > 0.5


def test_mesh_scene_deterministic_hit_ids(gpu_renderer):
    """Triangle meshes through the flattened BVH: bit-exact node AND face ids at 1 spp, pixel centre."""
    scene = Scene.load(scene_path("c2_mesh_room.xml"))
    gpu_renderer.LoadScene(scene)
    p = params(960, 540, spp=1, jitter=0, bounces=0)
    g = gpu_renderer.Render(p)
    c = oracle_ffi.OracleScene(scene).render(p)
    assert np.array_equal(g.hit_node, c.hit_node)
    assert np.array_equal(g.hit_face, c.hit_face)
    assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4
    assert (c.hit_face >= 0).mean() > 0.01


@pytest.mark.parametrize("name,spp", [("cornell_box.xml", 8), ("c2_mesh_room.xml", 4)])
def test_stochastic_same_seed(gpu_renderer, name, spp):
    """Reflection / refraction / soft shadows at the same Philox seed and spp.  Differences come only from
    transcendental ulps and float-sum order: gate of common.assert_stochastic_close (>= 99.95 % of the pixels within 1e-3,
    PSNR >= 80 dB, mean |diff| <= 1e-6)."""
    scene = Scene.load(scene_path(name))
    gpu_renderer.LoadScene(scene)
    p = params(480, 270, spp=spp, jitter=1, bounces=3)
    g = gpu_renderer.Render(p)
    c = oracle_ffi.OracleScene(scene).render(p)
    print(name, assert_stochastic_close(g, c, name))


def test_sample_ranges_sum_to_full_render(gpu_renderer):
    """Multi-GPU sample split: accumulating ranges [0,4) and [4,8) equals one 8-sample render."""
    import ctypes as C
    import torch
    scene = Scene.load(scene_path("cornell_box.xml"))
    gpu_renderer.LoadScene(scene)
    p = params(320, 180, spp=8, jitter=1, bounces=2)
    full = gpu_renderer.Render(p)
    acc = torch.zeros(180 * 320 * 8, dtype=torch.float32, device="cuda:0")
    for b in (0, 4):
        p.sample_begin, p.sample_count = b, 4
        gpu_renderer.render_accumulate(p, acc.data_ptr())
    torch.cuda.synchronize()
    p.sample_begin, p.sample_count = 0, 0
    split = gpu_renderer.resolve(p, acc.data_ptr())
    assert np.allclose(split.radiance[..., :3], full.radiance[..., :3], rtol=1e-5, atol=1e-6)
    assert (split.radiance[..., 3] == 8).all()


def test_tile_split_accumulate_equals_full_frame(gpu_renderer):
    """Multi-GPU tile split: ranks render 32x32 tiles r, r + G, ... (tile_begin / tile_stride) into their own
    accumulators; the sum of the accumulators is EXACTLY the full frame's (disjoint pixels, every pixel by one rank) — two and
    three virtual ranks on one GPU, a mesh scene with three bounces, ragged frame size (partial tiles)."""
    import torch
    scene = Scene.load(scene_path("c2_mesh_room.xml"))
    gpu_renderer.LoadScene(scene)
    W, H = 328, 200
    p = params(W, H, spp=4, jitter=1, bounces=3)
    full = gpu_renderer.Render(p)
    for G in (2, 3):
        accs = []
        for rank in range(G):
            acc = torch.zeros(W * H * 8, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            q = params(W, H, spp=4, jitter=1, bounces=3)
            q.tile_begin, q.tile_stride = rank, G
            gpu_renderer.render_accumulate(q, acc.data_ptr())
            torch.cuda.synchronize()
            accs.append(acc)
        n = torch.stack([a.view(H, W, 8)[..., 3] for a in accs])
        assert ((n > 0).sum(dim=0) == 1).all(), "a pixel was rendered by no rank or by two"
        total = torch.stack(accs).sum(dim=0)          # what ncclReduce(sum) leaves on the root
        torch.cuda.synchronize()
        img = gpu_renderer.resolve(p, total.data_ptr())
        assert (img.radiance[..., 3] == 4).all()
        # (float atomics into a path's radiance slot are unordered run to run: equal to the order of those sums)
        assert np.allclose(img.radiance[..., :3], full.radiance[..., :3], rtol=1e-5, atol=1e-6)
        assert (img.img == full.img).mean() > 0.9995


def test_geometry_buffers_match_oracle(gpu_renderer):
    """rt_framebuffer depth / normal / albedo (RenderImage::zbuffer, renderer.h:39; Denoise guides, denoiser.h:16-17)."""
    check_geometry_buffers(gpu_renderer, Scene.load(scene_path("c2_mesh_room.xml")), 480, 270)
    check_geometry_buffers(gpu_renderer, Scene.load(os.path.join(ROOT, "scenes", "c3_cornell_textured.xml")), 480, 270)


def test_instance_bvh_matches_scene_graph_walk(gpu_renderer, monkeypatch):
    check_instance_bvh(gpu_renderer, monkeypatch, scene_path)


def test_extended_materials(gpu_renderer):
    check_extended_materials(gpu_renderer)


def test_texture_filter(gpu_renderer):
    check_texture_filter(gpu_renderer)


def test_boundary_extras(gpu_renderer):
    check_boundary_extras(gpu_renderer)


def test_edge_cases(gpu_renderer):
    check_edge_cases(gpu_renderer)


def test_random_scene_graphs(gpu_renderer):
    check_random_scene_graphs(gpu_renderer, seeds=(1, 2, 3, 4, 5, 6), n_rays=20000, w=192, h=128)


def test_philox_known_answers_on_the_device(gpu_renderer):
    """the kernels' Philox4x32-10 and float mapping on the MI355X against the Random123 vectors and the oracle."""
    check_philox_on_device(gpu_renderer)
