"""Host layer (CPU, not the hot path): XML dialect semantics, PNG/OBJ IO, BVH builder invariants, procedural meshes."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import scene_path
from raytracer_amd import Scene, _ffi

REF_SCENES = "/root/reference/RayTracer/scenes"

MINI = """<?xml version="1.0"?>
<xml>
  <!-- comment -->
  <scene>
    <Object type="Sphere" name="a" material="m1">
      <scale x="2" value="3"/> <rotate angle="90" z="1"/> <translate x="1" y="2" z="3"/>
      <object type="plane" name="b"/>
    </Object>
    <object name="group"/>
    <light type="Point" name="l"> <intensity value="2" r="0" g="1" b="0"/> <position x="1" value="2"/> <size value="0.5"/> </light>
    <light type="direct"> <intensity value="0.5"/> <direction x="0" y="3" z="-4"/> </light>
    <light type="ambient"> <intensity r="0.1"/> </light>
    <material type="blinn" name="m1">
      <diffuse value="0.7" r="1" g="0.5" b="0.25" texture="checkerboard">
        <color1 r="1" g="0" b="0"/> <color2 value="0.5" texture="checkerboard"> <scale value="0.25"/> </color2>
        <scale x="0.5" y="0.5"/>
      </diffuse>
      <refraction value="0.8" index="1.52"/> <absorption r="0.01" g="0.001" b="0.0001"/>
    </material>
    <material type="phong" name="m2"/>
    <background r="0.1" g="0.2" b="0.3"/>
    <unknowntag/>
  </scene>
  <camera gamma="sRGB"> <position x="0" y="-10" z="0"/> <target x="0" y="0" z="0"/> <up x="0" y="1" z="1"/>
    <fov value="30"/> <width value="64"/> <height value="48"/> </camera>
</xml>"""


def test_xml_dialect_semantics():
    s = Scene.from_string(MINI)
    d = s.desc
    assert d.n_nodes == 4 and s.warnings == 1  # root + a + b + group; one unknown tag
    a, b, g = d.nodes[1], d.nodes[2], d.nodes[3]
    assert (a.parent, a.prim_kind, a.material_id) == (0, 1, 0)
    assert (b.parent, b.prim_kind, b.material_id) == (1, 2, -1)
    assert (g.parent, g.prim_kind) == (0, 0)
    # scale(6,3,3) then rotate 90 about z then translate: tm = T * R * S  (document order, scene.h:165)
    tm = np.array(a.tm.m[:], np.float64).reshape(4, 3).T  # columns
    assert np.allclose(tm[:, 3], [1, 2, 3])
    assert np.allclose(tm[:, 0], [0, 6, 0], atol=1e-6) and np.allclose(tm[:, 1], [-3, 0, 0], atol=1e-6) and np.allclose(tm[:, 2], [0, 0, 3], atol=1e-6)
    itm = np.array(a.itm.m[:], np.float64).reshape(4, 3).T
    full = np.vstack([tm, [0, 0, 0, 1]]) @ np.vstack([itm, [0, 0, 0, 1]])
    assert np.allclose(full, np.eye(4), atol=1e-5)
    # lights: value multiplies the per-channel attributes (xmlload.h:62-63)
    l0, l1, l2 = d.lights[0], d.lights[1], d.lights[2]
    assert l0.kind == 3 and tuple(l0.intensity) == (0.0, 2.0, 0.0) and tuple(l0.position) == (2.0, 0.0, 0.0) and l0.size == 0.5
    assert l1.kind == 2 and np.allclose(l1.direction[:], [0, 0.6, -0.8])
    assert l2.kind == 1 and np.allclose(l2.intensity[:], [0.1, 1, 1])
    # material: defaults kept for absent children (materials.h:53-60), nested checkerboard
    m = d.materials[0]
    assert m.kind == 1 and np.allclose(m.diffuse.value[:], [0.7, 0.35, 0.175]) and m.diffuse.map >= 0
    assert np.allclose(m.specular.value[:], [0.7] * 3) and m.glossiness.value == 20.0
    assert np.allclose(m.refraction.value[:], [0.8] * 3) and abs(m.ior - 1.52) < 1e-6
    assert np.allclose(m.absorption[:], [0.01, 0.001, 0.0001])
    outer = d.texmaps[m.diffuse.map]
    assert outer.kind == 1 and np.allclose(outer.color[0].value[:], [1, 0, 0]) and outer.color[0].map == -1
    inner = d.texmaps[outer.color[1].map]
    assert inner.kind == 1 and np.allclose(outer.color[1].value[:], [0.5] * 3)
    assert np.allclose(np.array(inner.itm.m[:])[[0, 4, 8]], [4, 4, 4])       # inverse of scale 0.25
    assert np.allclose(np.array(outer.itm.m[:])[[0, 4, 8]], [2, 2, 1])
    assert d.materials[1].kind == 3  # RT_MTL_PHONG: shades white (the reference's stub) unless RT_EXT_MATERIALS is set
    assert np.allclose(d.background.value[:], [0.1, 0.2, 0.3]) and d.background.map == -1
    cam = d.camera
    assert cam.srgb == 1 and (cam.width, cam.height) == (64, 48) and cam.fov == 30
    assert np.allclose(cam.dir[:], [0, 1, 0]) and np.allclose(cam.up[:], [0, 0, 1])  # up re-orthogonalised (xmlload.cpp:118-119)


def test_load_errors():
    with pytest.raises(RuntimeError):
        Scene.from_string("<xml><scene></scene></xml>")  # no camera
    with pytest.raises(RuntimeError):
        Scene.from_string("<xml><scene><object></scene></xml>")  # malformed
    with pytest.raises(RuntimeError):
        Scene.load("/nonexistent/scene.xml")


def test_png_roundtrip(tmp_path):
    host = _ffi.load_host()
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "a.png").encode()
    assert host.rth_write_png(p, img.ctypes.data, 53, 37, 3) == 0
    ptr = C.POINTER(C.c_uint8)()
    w, h = C.c_int(), C.c_int()
    assert host.rth_read_image_rgb(p, C.byref(ptr), C.byref(w), C.byref(h)) == 0
    back = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    host.rth_free(ptr)
    assert (w.value, h.value) == (53, 37) and np.array_equal(back, img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p.decode()).convert("RGB")), img)
    # a palettised PNG written by PIL decodes to the same RGB
    pal = Image.fromarray(img).quantize(16)
    pal.save(str(tmp_path / "p.png"))
    assert host.rth_read_image_rgb(str(tmp_path / "p.png").encode(), C.byref(ptr), C.byref(w), C.byref(h)) == 0
    back = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    host.rth_free(ptr)
    assert np.array_equal(back, np.asarray(pal.convert("RGB")))


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference checkout not present")
def test_every_bundled_texture_decodes_like_lct_rgb():
    """TextureFile::Load decodes PNGs with lodepng to LCT_RGB, 8 bit (texture.cpp:86).  The reference's textures are
    palettised at 2 / 4 / 8 bits, RGBA, grey and 1-bit grey; each must decode to the RGB an independent decoder gives."""
    from PIL import Image
    host = _ffi.load_host()
    files = sorted(glob.glob(os.path.join(REF_SCENES, "**", "*.png"), recursive=True))
    assert len(files) > 100
    kinds = set()
    for f in files:
        ptr = C.POINTER(C.c_uint8)()
        w, h = C.c_int(), C.c_int()
        assert host.rth_read_image_rgb(f.encode(), C.byref(ptr), C.byref(w), C.byref(h)) == 0, f
        got = np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
        host.rth_free(ptr)
        im = Image.open(f)
        kinds.add(im.mode)
        ref = np.asarray(im.convert("RGBA").convert("RGB") if im.mode in ("P", "RGBA", "LA", "PA") else im.convert("RGB"))
        assert got.shape == ref.shape and np.array_equal(got, ref), f
    assert {"P", "RGBA"} <= kinds


def _check_bvh(mesh):
    nf = mesh.n_faces
    pos = np.ctypeslib.as_array(mesh.positions, shape=(mesh.n_positions, 3))
    fidx = np.ctypeslib.as_array(mesh.pos_idx, shape=(nf, 3))
    elems = np.ctypeslib.as_array(mesh.bvh_elements, shape=(mesh.n_bvh_elements,))
    assert sorted(elems.tolist()) == list(range(nf)), "every face in exactly one leaf"
    seen = 0
    stack = [(0, 0)]
    max_depth = 0
    while stack:
        i, depth = stack.pop()
        n = mesh.bvh[i]
        max_depth = max(max_depth, depth)
        lo, hi = np.array(n.bmin[:]), np.array(n.bmax[:])
        if n.b & 0x80000000:
            cnt = n.b & 0x7fffffff
            assert 1 <= cnt <= 4
            for f in elems[n.a:n.a + cnt]:
                v = pos[fidx[f]]
                assert (v >= lo - 1e-6).all() and (v <= hi + 1e-6).all()
            seen += cnt
        else:
            for c in (n.a, n.b):
                cn = mesh.bvh[c]
                assert (np.array(cn.bmin[:]) >= lo - 1e-6).all() and (np.array(cn.bmax[:]) <= hi + 1e-6).all()
                stack.append((c, depth + 1))
    assert seen == nf and max_depth <= 32
    return max_depth


def test_procedural_mesh_and_bvh():
    s = Scene.load(scene_path("c2_mesh_room.xml"))
    assert s.desc.n_meshes == 2 and s.warnings == 0
    sizes = [s.desc.meshes[i].n_faces for i in range(2)]
    assert 55000 < sizes[0] < 70000 and 13000 < sizes[1] < 18000
    for i in range(2):
        _check_bvh(s.desc.meshes[i])
    # seeded: loading twice gives identical vertices
    s2 = Scene.load(scene_path("c2_mesh_room.xml"))
    m1, m2 = s.desc.meshes[0], s2.desc.meshes[0]
    a = np.ctypeslib.as_array(m1.positions, shape=(m1.n_positions * 3,))
    b = np.ctypeslib.as_array(m2.positions, shape=(m2.n_positions * 3,))
    assert np.array_equal(a, b)


def test_obj_reader(tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("# quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nf 1/1 2/2 3/3 4/4\nf -4 -3 -2\n")
    xml = f"""<xml><scene><object type="obj" name="{obj}" /></scene><camera/></xml>"""
    s = Scene.from_string(xml)
    m = s.desc.meshes[0]
    assert m.n_faces == 3 and m.n_positions == 4 and m.n_normals == 4  # fan triangulation, normals computed (objects.h:63)
    nrm = np.ctypeslib.as_array(m.normals, shape=(4, 3))
    assert np.allclose(nrm, [[0, 0, 1]] * 4)
    _check_bvh(m)


def test_multimtl_from_obj_material_library():
    """An <object type="obj"> without material= whose OBJ names a material library gets a MultiMtl built from it
    (LoadNode xmlload.cpp:159, SetNodeMaterials :294-315, CreateMultiMtl :319-343): Kd / Ks / Ns / Ni, map_Kd as the diffuse
    texture, illum 3..7 => reflection = Ks, illum >= 6 => refraction = 1 - Tf, map_Ks replacing the DIFFUSE texture (sic,
    :331) and becoming the reflection texture.  The triangle test never sets mtlID, so sub-material 0 — the
    library's FIRST newmtl, not the group a face belongs to — is what shades; the flattened scene binds it directly."""
    from common import load_multimtl_scene
    s = load_multimtl_scene()
    d = s.desc
    assert s.warnings == 0 and d.n_meshes == 4 and d.n_nodes == 8
    names = [s.node_name(i) for i in range(d.n_nodes)]
    mat = lambda i: d.materials[d.nodes[i].material_id]
    # every mesh is the lathe body (24 x 12 quads) + the canopy quad, fan-triangulated, with its own uvs
    for i in range(4):
        assert d.meshes[i].n_faces == 2 * 24 * 12 + 2 and d.meshes[i].n_uvw == 25 * 13 + 4 and d.meshes[i].uvw_idx
    m = mat(names.index("tests/golden/meshes/mm_illum2.obj"))      # illum 2: no reflection, no refraction, map_Kd
    assert m.kind == 1 and np.allclose(m.diffuse.value[:], [0.8, 0.6, 0.4]) and np.allclose(m.specular.value[:], [0.3] * 3)
    assert m.glossiness.value == 40.0 and abs(m.ior - 1.3) < 1e-6
    assert np.allclose(m.reflection.value[:], 0) and np.allclose(m.refraction.value[:], 0)
    tm = d.texmaps[m.diffuse.map]
    assert tm.kind == 2 and d.images[tm.image].width == 1024 and np.allclose(np.array(tm.itm.m[:])[[0, 4, 8]], 1)
    first3 = names.index("tests/golden/meshes/mm_illum3.obj")
    m = mat(first3)                                                  # illum 3: reflection = Ks
    assert np.allclose(m.diffuse.value[:], [0.3, 0.5, 0.7]) and np.allclose(m.reflection.value[:], [0.5, 0.4, 0.3])
    assert np.allclose(m.refraction.value[:], 0) and m.glossiness.value == 120.0 and abs(m.ior - 1.1) < 1e-6 and m.diffuse.map == -1
    m = mat(names.index("tests/golden/meshes/mm_illum6.obj"))      # illum 6: reflection = Ks, refraction = 1 - Tf
    assert np.allclose(m.reflection.value[:], [0.6] * 3) and np.allclose(m.refraction.value[:], [0.85, 0.9, 0.95])
    assert abs(m.ior - 1.45) < 1e-6 and m.glossiness.value == 80.0
    # a material already named like the node wins over the library (materials.Find(node->GetName()), :300)
    m = mat(names.index("tests/golden/meshes/mm_mapks.obj"))
    assert np.allclose(m.diffuse.value[:], [0.9, 0.1, 0.9]) and m.diffuse.map == -1
    # the second node of the same file finds the material the first one created; material= beats the library
    second3 = len(names) - 1 - names[::-1].index("tests/golden/meshes/mm_illum3.obj")
    assert second3 != first3 and d.nodes[second3].material_id == d.nodes[first3].material_id
    assert d.nodes[second3].mesh_id == d.nodes[first3].mesh_id          # objList.Find(name), :146
    last6 = len(names) - 1 - names[::-1].index("tests/golden/meshes/mm_illum6.obj")
    assert d.materials[d.nodes[last6].material_id].diffuse.value[0] == np.float32(0.6)
    # the map_Ks quirk, on a scene where no material of that name pre-exists
    from conftest import ROOT
    q = Scene.from_string("""<xml><scene><object type="obj" name="tests/golden/meshes/mm_mapks.obj"/></scene><camera/></xml>""", asset_dir=ROOT)
    m = q.desc.materials[q.desc.nodes[1].material_id]
    dm, rm = q.desc.texmaps[m.diffuse.map], q.desc.texmaps[m.reflection.map]
    assert q.desc.images[dm.image].width == 160 and q.desc.images[rm.image].width == 160     # map_Ks (block2.png), not map_Kd
    assert np.allclose(m.reflection.value[:], [0.4] * 3) and np.allclose(m.refraction.value[:], 0)  # illum 4
    # an OBJ without a library on a node without material=: no material at all (null material)
    q = Scene.from_string("""<xml><scene><object type="obj" name="gen:icosphere?seed=1&amp;tris=80"/></scene><camera/></xml>""")
    assert q.desc.nodes[1].material_id == -1 and q.desc.n_materials == 0


def test_stress_scene_generator():
    s = Scene.stress(seed=7, n_instances=16, subdiv=2, ground_res=16, width=64, height=36)
    m = s.desc.meshes[0]
    assert m.n_faces == 16 * 320 + 2 * 16 * 16
    _check_bvh(m)


@pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="reference checkout not present on this machine")
def test_every_bundled_reference_scene_loads():
    """north_star: every bundled scene loads through the re-authored loader (missing .obj assets only warn)."""
    files = sorted(glob.glob(os.path.join(REF_SCENES, "*.xml"))) + [os.path.join(REF_SCENES, "minecraft_world", "mc_world.xml")]
    assert len(files) >= 17
    for f in files:
        s = Scene.load(f)
        assert s.desc.n_nodes >= 2 and s.desc.camera.width > 0
    s = Scene.load(os.path.join(REF_SCENES, "volumetric_spiral.xml"))
    assert s.desc.n_nodes == 2947
    s = Scene.load(os.path.join(REF_SCENES, "scene_textures.xml"))
    assert s.desc.n_images == 2 and s.desc.images[0].width == 1024


def test_malformed_png_files_are_refused(tmp_path):
    """a scene file can name any texture file — truncated headers, impossible bit depths and absurd sizes must
    fail the load (return -1), not read out of bounds, divide by zero or throw through the C boundary."""
    import ctypes as C
    import struct
    import zlib
    from raytracer_amd import _ffi
    host = _ffi.load_host()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    def png(ihdr, idat=b""):
        return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")

    good = png(struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 0), zlib.compress(b"\x00" + b"\x10" * 6 + b"\x00" + b"\x20" * 6))
    cases = {
        "good": (good, 0),
        "short_ihdr": (png(struct.pack(">II", 2, 2)), -1),
        "depth0": (png(struct.pack(">IIBBBBB", 2, 2, 0, 0, 0, 0, 0), zlib.compress(b"\x00\x00" * 2)), -1),
        "depth3": (png(struct.pack(">IIBBBBB", 2, 2, 3, 0, 0, 0, 0), zlib.compress(b"\x00\x00" * 2)), -1),
        "depth32": (png(struct.pack(">IIBBBBB", 2, 2, 32, 2, 0, 0, 0), zlib.compress(b"\x00" * 50)), -1),
        "huge": (png(struct.pack(">IIBBBBB", 1 << 20, 1 << 20, 8, 2, 0, 0, 0), zlib.compress(b"\x00" * 16)), -1),
        "truncated": (good[:40], -1),
    }
    for name, (data, want) in cases.items():
        f = tmp_path / (name + ".png")
        f.write_bytes(data)
        rgb, w, h = C.POINTER(C.c_uint8)(), C.c_int(), C.c_int()
        rc = host.rth_read_image_rgb(str(f).encode(), C.byref(rgb), C.byref(w), C.byref(h))
        assert rc == want, name
        if rc == 0:
            assert (w.value, h.value) == (2, 2) and rgb[0] == 0x10 and rgb[6] == 0x20
            host.rth_free(rgb)
