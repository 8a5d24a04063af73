"""CPU-side check of the photon pass and gather logic (host emulation build vs the oracle); see photon_checks.py."""
import pytest

import emu_ffi
import photon_checks as pc
from conftest import scene_path
from raytracer_amd import Scene


@pytest.fixture(scope="module")
def emu():
    return emu_ffi.EmuRenderer()


def test_gather_matches_kdtree_walk(emu):
    pc.check_gather_matches_kdtree_walk(emu, Scene.load(scene_path("cornell_box.xml")), n_photons=3000, n_queries=200)


def test_photon_maps_match_sequential_emitter(emu):
    n = pc.check_photon_maps_match_sequential_emitter(emu, Scene.load(scene_path("cornell_box.xml")), n_photons=1500)
    assert n[0] == 1500 and n[1] == 1500


def test_emit_limit_leaves_caustics_map_partly_filled(emu):
    """The reference loops forever when a map cannot fill (RayTracer.cpp:277); photon_emit_limit bounds it."""
    n = pc.check_photon_maps_match_sequential_emitter(emu, Scene.load(scene_path("cornell_box.xml")), n_photons=1500,
                                                      emit_limit=2500)
    assert n[0] == 1500 and 0 < n[1] < 1500


DIFFUSE_ONLY = """<xml><scene>
  <object type="plane" name="floor" material="m"> <scale value="20"/> </object>
  <object type="sphere" name="ball" material="m"> <scale value="2"/> <translate z="2"/> </object>
  <material type="blinn" name="m"> <diffuse r="0.8" g="0.6" b="0.4"/> <specular value="0"/> </material>
  <light type="point"> <intensity value="50"/> <position x="1" y="-2" z="9"/> <size value="0.5"/> </light>
</scene><camera> <position x="0" y="-20" z="6"/> <target x="0" y="0" z="2"/> <up x="0" y="0" z="1"/> <fov value="40"/>
  <width value="64"/> <height value="48"/> </camera></xml>"""


def test_no_specular_path_scene_terminates(emu):
    """No specular -> diffuse path exists, so the caustics map stays empty; the reference would never return."""
    n = pc.check_photon_maps_match_sequential_emitter(emu, Scene.from_string(DIFFUSE_ONLY), n_photons=300)
    assert n[0] == 300 and n[1] == 0


def test_photon_render(emu):
    pc.check_photon_render(emu, Scene.load(scene_path("cornell_box.xml")), n_photons=2000, w=48, h=27, spp=2, tol=1e-4)


def test_gather_hints_change_walks_only(emu):
    pc.check_gather_hints(emu, Scene.load(scene_path("cornell_box.xml")), n_photons=4000, w=48, h=27, spp=2)


def test_gather_bracket_search_growth_and_ties(emu):
    pc.check_gather_bracket_search(emu, Scene.load(scene_path("cornell_box.xml")))


def test_kd_tree_build_matches_reference_balance(emu):
    pc.check_kd_tree_build(emu, Scene.load(scene_path("cornell_box.xml")))


def test_c4_adaptive_photons_denoise_end_to_end(emu):
    """BASELINE config 4's pipeline at emulation size (single-threaded kernels): adaptive sampling WITH the photon term, both
    maps, then the denoise pass.  The GPU test runs it with the reference's constants (100 000 photons, k = 128, 32..128)."""
    s = pc.check_c4_as_defined(emu, Scene.load(scene_path("cornell_box.xml")), 32, 18, n_photons=3000, k=32, spp=(4, 16), threshold=0.03,
                               equal_frac=1.0, tol_mean=2e-5, psnr_min=60.0)
    print(s)
