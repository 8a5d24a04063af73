"""GPU parity: batched TraceRay / TraceShadowRay through the C-ABI against the CPU oracle (bit-exact decisions)."""
import numpy as np
import pytest

import oracle_ffi
from common import assert_hits_equal, random_rays
from conftest import scene_path
from raytracer_amd import RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK, Scene

pytestmark = pytest.mark.gpu

SCENES = ["spheres_room.xml", "cornell_box.xml", "c2_mesh_room.xml"]


@pytest.mark.parametrize("name", SCENES)
@pytest.mark.parametrize("side", [RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK])
def test_trace_closest_matches_oracle(gpu_renderer, name, side):
    scene = Scene.load(scene_path(name))
    gpu_renderer.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    rays = random_rays(20000, seed=11)
    assert_hits_equal(gpu_renderer.TraceRay(rays, side), orc.trace_closest(rays, side), f"{name} side={side}")


@pytest.mark.parametrize("name", SCENES)
def test_trace_any_matches_oracle(gpu_renderer, name):
    scene = Scene.load(scene_path(name))
    gpu_renderer.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    rays = random_rays(20000, seed=12)
    rng = np.random.default_rng(5)
    tmax = (0.2 + 2.0 * rng.random(len(rays))).astype(np.float32)
    tmax[::7] = np.finfo(np.float32).max
    g = gpu_renderer.TraceShadowRay(rays, tmax)
    c = orc.trace_any(rays, tmax)
    assert np.array_equal(g, c), f"{name}: {np.count_nonzero(g != c)} occlusion results differ"
    assert 0.05 < c.mean() < 0.999


def test_empty_batch_and_errors(gpu_renderer):
    scene = Scene.load(scene_path("spheres_room.xml"))
    gpu_renderer.LoadScene(scene)
    assert len(gpu_renderer.TraceRay(random_rays(0, 1))) == 0
