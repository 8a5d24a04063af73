"""GPU parity of the photon pass (k_photon_trace) and the irradiance gather (k_photon_gather) through the C-ABI."""
import numpy as np
import pytest

import photon_checks as pc
from conftest import scene_path
from raytracer_amd import Scene

pytestmark = pytest.mark.gpu


def test_gather_matches_reference_heap(gpu_renderer):
    pc.check_gather_matches_kdtree_walk(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), n_photons=20000,
                                        n_queries=4096)


def test_kd_tree_build_matches_reference_balance(gpu_renderer):
    pc.check_kd_tree_build(gpu_renderer, Scene.load(scene_path("cornell_box.xml")))


def test_gather_bracket_search_growth_and_ties(gpu_renderer):
    pc.check_gather_bracket_search(gpu_renderer, Scene.load(scene_path("cornell_box.xml")))


def test_photon_maps_match_sequential_emitter(gpu_renderer):
    n = pc.check_photon_maps_match_sequential_emitter(gpu_renderer, Scene.load(scene_path("cornell_box.xml")),
                                                      n_photons=20000, exact=False)
    assert n[0] == 20000 and n[1] == 20000
    print(n)


def test_photon_maps_reference_size(gpu_renderer):
    """numPhotons = 100000 per map (raytracer.h:23): both maps fill, the build is reproducible run to run."""
    scene = Scene.load(scene_path("cornell_box.xml"))
    gpu_renderer.LoadScene(scene)
    p = pc.photon_params(100000)
    gpu_renderer.BuildPhotonMaps(p)
    a0, a1 = gpu_renderer.GetPhotonMap(), gpu_renderer.GetCausticsMap()
    gpu_renderer.BuildPhotonMaps(p)
    b0, b1 = gpu_renderer.GetPhotonMap(), gpu_renderer.GetCausticsMap()
    assert len(a0) == 100000 and len(a1) == 100000
    assert a0.tobytes() == b0.tobytes() and a1.tobytes() == b1.tobytes()


def test_photon_render(gpu_renderer):
    """Gate = what the device achieves with margin (measured: > 90 dB, mean |diff| < 1e-6): PSNR >= 80 dB, mean <= 1e-5."""
    pc.check_photon_render(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), n_photons=20000, w=240, h=135, spp=4,
                           psnr_min=80.0, tol=1e-5)


def test_c4_adaptive_photons_denoise_end_to_end(gpu_renderer):
    """BASELINE config 4 as defined, with the reference's constants (2 x 100 000 photons, k = 128, r = 3, e = 1 / 0.25, adaptive
    32..128 at 0.01, three bounces, 16..128 shadow rays, then the denoise pass), against the oracle gathering from the
    device's trees: sample counts equal for >= 99.95 % of the pixels, radiance and the denoised 8-bit frame compared."""
    s = pc.check_c4_as_defined(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), 160, 90)
    print("c4 as defined:", s)


def test_gather_hints_change_walks_only(gpu_renderer):
    pc.check_gather_hints(gpu_renderer, Scene.load(scene_path("cornell_box.xml")), n_photons=20000, w=240, h=135, spp=4, tol=1e-5)


def test_undersized_queues_with_photons_and_soft_shadows():
    """Every level of a pass appends its shadow groups and its photon lookups to one set of queues, served once after the last
    level.  With the measured queue sizing shrunk 20x the pass overflows, leaves the accumulator alone and is redone in halves:
    the frame must equal the one rendered with queues of the measured size (same library, same photon maps; photon frames
    agree to the order of their float sums), and the work of the overflowed attempts must not be counted."""
    from raytracer_amd import Renderer
    scene = Scene.load(scene_path("cornell_box.xml"))
    p = pc.photon_params(20000, w=192, h=108, spp=4, bounces=3, k=32)
    p.shadow_min, p.shadow_max = 16, 128
    p.frame_mem_mb = 1   # the smallest pass size (65536 camera samples): the queues are then sized for the passes of this frame
    frames, counts, launches = [], [], []
    for fan_scale in (1.0, 0.05):
        r = Renderer(0)
        r.set_option("fan_scale", fan_scale)
        r.LoadScene(scene)
        r.BuildPhotonMaps(p)
        r.counters(enable=True, reset=True)
        r.profile(enable=True, reset=True)
        frames.append(r.Render(p))
        counts.append(r.counters())
        launches.append(r.profile()["raygen"]["launches"])
    assert launches[1] > launches[0], "the pass was not split: the test did not reach the overflow path"
    a, b = frames[0].radiance, frames[1].radiance
    assert (a[..., 3] == 4).all() and (b[..., 3] == 4).all()
    rel = np.abs(a[..., :3] - b[..., :3]) / np.maximum(np.abs(a[..., :3]), 1e-3)
    assert rel.max() <= 1e-5, rel.max()   # (measured: 3e-7)
    for k in ("rays_primary", "rays_secondary", "paths", "shades", "gathers"):
        assert counts[0][k] == counts[1][k], (k, counts[0][k], counts[1][k])
