"""The oracle's timing-only mode (orc_set_accel: instance BVH + ordered, pruned mesh BVH walks — bench.py's second cpu_baseline
row, SURVEY 8d "same data structures as GPU") finds the hits of the reference-structure walk: same occlusion, same closest hit
up to exact-t ties between different instances, and is the faster of the two on a sphere-heavy scene."""
import time

import numpy as np

import oracle_ffi
from common import params, random_rays, random_scene_xml
from conftest import scene_path
from raytracer_amd import Scene


def _compare(scene, rays, label):
    orc = oracle_ffi.OracleScene(scene)
    tmax = np.full(len(rays), 5.0, np.float32)
    tmax[::3] = np.finfo(np.float32).max
    ref_hits, ref_occ = orc.trace_closest(rays, 3), orc.trace_any(rays, tmax)
    orc.set_accel(1)
    acc_hits, acc_occ = orc.trace_closest(rays, 3), orc.trace_any(rays, tmax)
    orc.set_accel(0)
    assert np.array_equal(ref_occ, acc_occ), label
    assert np.array_equal(ref_hits["z"], acc_hits["z"]), label            # the nearest distance never depends on the walk
    same = (ref_hits["node"] == acc_hits["node"]) & (ref_hits["face"] == acc_hits["face"])
    assert same.mean() >= 0.995, (label, same.mean())                      # (exact-t ties between coincident instances)
    assert np.array_equal(orc.trace_closest(rays, 3)["node"], ref_hits["node"]), "set_accel(0) restores the reference walk"
    return orc


def test_accel_walk_finds_the_same_hits():
    for name in ("spheres_room.xml", "c2_mesh_room.xml", "ref_finalProject.xml"):
        scene = Scene.load(scene_path(name))
        rays = random_rays(4000, seed=5, origin_box=((-20, -30, 2), (20, -5, 25)), target_box=((-8, -8, 0), (8, 8, 12)))
        _compare(scene, rays, name)
    for seed in (3, 4):
        scene = Scene.from_string(random_scene_xml(seed))
        rays = random_rays(4000, seed=100 + seed, origin_box=((-20, -30, -8), (20, 20, 16)), target_box=((-8, -8, -6), (8, 8, 8)))
        _compare(scene, rays, f"random {seed}")


def test_accel_frame_and_speed_on_a_sphere_field():
    scene = Scene.load(scene_path("ref_volumetric_spiral.xml"))
    orc = oracle_ffi.OracleScene(scene)
    p = params(96, 54, spp=1, jitter=0, bounces=1)
    t0 = time.perf_counter()
    ref = orc.render(p, 4)
    t_ref = time.perf_counter() - t0
    orc.set_accel(1)
    t0 = time.perf_counter()
    acc = orc.render(p, 4)
    t_acc = time.perf_counter() - t0
    assert (ref.hit_node == acc.hit_node).mean() >= 0.999
    assert np.abs(ref.radiance - acc.radiance).mean() <= 1e-4
    assert t_acc < t_ref, (t_acc, t_ref)   # Theta(nodes) per ray against a tree walk: 2945 spheres
