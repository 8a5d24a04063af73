"""Shared helpers for the parity tests."""
import ctypes as C

import numpy as np

from raytracer_amd import default_params, rays_from_arrays


def params(width, height, spp=1, jitter=0, bounces=0, photons=0, light_scale=1.0, mc=0, seed=1):
    p = default_params()
    p.width, p.height = width, height
    p.spp_min = p.spp_max = spp
    p.pixel_jitter = jitter
    p.max_bounces = bounces
    p.photons = photons
    p.light_size_scale = light_scale
    p.mc_bounces = mc
    p.seed_lo = seed
    return p


def rel_colour_diff(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor) over rgb"""
    a = a[..., :3].astype(np.float64)
    b = b[..., :3].astype(np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def psnr8(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float((d * d).mean())
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)


def random_rays(n, seed, origin_box=((-20, -70, -2), (20, 30, 26)), target_box=((-16, -12, -1), (16, 21, 25))):
    """Seeded rays from points in origin_box towards points in target_box (unnormalised directions)."""
    rng = np.random.default_rng(seed)
    lo, hi = np.array(origin_box[0], np.float32), np.array(origin_box[1], np.float32)
    o = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    lo, hi = np.array(target_box[0], np.float32), np.array(target_box[1], np.float32)
    t = (lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)).astype(np.float32)
    d = (t - o).astype(np.float32)
    scale = (0.25 + 2.0 * rng.random((n, 1), dtype=np.float32)).astype(np.float32)
    return rays_from_arrays(o, d * scale)


def assert_hits_equal(gpu, cpu, what=""):
    """Bit-exact on decisions (node / face / light / front / miss) and on z; p and N to 1e-5 relative, uvw to 1e-5 abs."""
    for k in ("node", "face", "light"):
        assert np.array_equal(gpu[k], cpu[k]), f"{what}: {k} differs at {np.flatnonzero(gpu[k] != cpu[k])[:8]}"
    hit = (cpu["node"] >= 0) | (cpu["light"] >= 0)
    assert np.array_equal(gpu["front"][hit], cpu["front"][hit]), f"{what}: front flag differs"
    assert np.array_equal(gpu["z"], cpu["z"]), f"{what}: z differs (max {np.abs(gpu['z'][hit] - cpu['z'][hit]).max()})"
    surf = cpu["node"] >= 0
    for k in ("p", "N"):
        d = np.abs(gpu[k][surf] - cpu[k][surf])
        s = np.maximum(np.abs(cpu[k][surf]), 1.0)
        assert (d / s).max(initial=0) <= 1e-5, f"{what}: {k} differs by {(d / s).max()}"
    d = np.abs(gpu["uvw"][surf] - cpu["uvw"][surf])
    assert d.max(initial=0) <= 1e-5, f"{what}: uvw differs by {d.max()}"


def check_geometry_buffers(r, scene, w, h):
    """depth / normal / albedo of the first camera sample's primary hit against the oracle."""
    import oracle_ffi
    r.LoadScene(scene)
    p = params(w, h, spp=2, jitter=1, bounces=1)
    g = r.Render(p)
    c = oracle_ffi.OracleScene(scene).render(p)
    assert np.array_equal(g.hit_node, c.hit_node)
    hit = np.isfinite(c.zbuffer) & (c.zbuffer < 1e30)
    assert hit.mean() > 0.3 and (~hit).sum() == (g.zbuffer >= 1e30).sum()
    assert np.allclose(g.zbuffer[hit], c.zbuffer[hit], rtol=1e-6)
    assert np.array_equal(g.zbuffer[~hit], c.zbuffer[~hit])
    assert np.abs(g.normal - c.normal).max() <= 1e-5
    assert np.abs(g.albedo - c.albedo).max() <= 1e-6
    surf = c.hit_node >= 0
    assert np.allclose(np.linalg.norm(c.normal[surf], axis=1), 1.0, atol=1e-5)


def check_instance_bvh(r, monkeypatch, scene_path_fn):
    """The instance BVH (TLAS) against the oracle's Theta(nodes) scene-graph walk (RayTracer.cpp:373-435): a field of
    overlapping spheres (equal-t ties are decided by depth-first order) and, with the threshold forced down, the
    stock scenes including mesh instances — frames through the two-phase kernels, batched queries through the
    one-phase walk with its nested mesh walk."""
    import sys
    import oracle_ffi
    from raytracer_amd import RT_HIT_FRONT_AND_BACK, Scene
    import bench
    field = Scene.from_string(bench.sphere_field_xml(n=400, seed=3, width=192, height=108))
    r.LoadScene(field)
    orc = oracle_ffi.OracleScene(field)
    p = params(192, 108, spp=1, jitter=0, bounces=0)
    g, c = r.Render(p), orc.render(p)
    assert np.array_equal(g.hit_node, c.hit_node) and rel_colour_diff(g.radiance, c.radiance) <= 1e-4
    assert len(np.unique(c.hit_node)) > 40
    p = params(96, 54, spp=2, jitter=1, bounces=3)
    g, c = r.Render(p), orc.render(p)
    assert_stochastic_close(g, c, "sphere field")
    rays = random_rays(4000, seed=21, origin_box=((-60, -40, 0), (60, 380, 80)), target_box=((-50, 0, 0), (50, 360, 60)))
    assert_hits_equal(r.TraceRay(rays, RT_HIT_FRONT_AND_BACK), orc.trace_closest(rays, RT_HIT_FRONT_AND_BACK), "sphere field")
    tmax = np.full(len(rays), 0.8, np.float32)
    assert np.array_equal(r.TraceShadowRay(rays, tmax), orc.trace_any(rays, tmax))
    for name in ("spheres_room.xml", "cornell_box.xml", "c2_mesh_room.xml"):
        scene = Scene.load(scene_path_fn(name))
        r.set_option("tlas_min", 2)
        try:
            r.LoadScene(scene)
        finally:
            r.set_option("tlas_min", 24)
        orc = oracle_ffi.OracleScene(scene)
        p = params(160, 90, spp=1, jitter=0, bounces=0)
        g, c = r.Render(p), orc.render(p)
        assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face), name
        assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4, name
        rays = random_rays(3000, seed=31)
        assert_hits_equal(r.TraceRay(rays, RT_HIT_FRONT_AND_BACK), orc.trace_closest(rays, RT_HIT_FRONT_AND_BACK), name)
        tmax = np.full(len(rays), 1.5, np.float32)
        assert np.array_equal(r.TraceShadowRay(rays, tmax), orc.trace_any(rays, tmax)), name
        p = params(96, 54, spp=2, jitter=1, bounces=2)
        g, c = r.Render(p), orc.render(p)
        assert_stochastic_close(g, c, name)


TEXTURED = """<xml><scene>
  <object type="plane" name="floor" material="chk"> <scale value="20"/> </object>
  <object type="sphere" name="ball" material="glass"> <scale value="3"/> <translate x="-4" z="3"/> </object>
  <object type="sphere" name="mirror" material="mir"> <scale value="3"/> <translate x="4" z="3"/> </object>
  <object type="sphere" name="white" material="wht"> <translate x="0" y="-4" z="1"/> </object>
  <material type="blinn" name="chk">
    <diffuse texture="checkerboard"> <color1 r="0.9" g="0.2" b="0.2"/>
      <color2 value="0.8" texture="checkerboard"> <color1 r="0.1" g="0.1" b="0.8"/> <scale value="0.125"/> </color2>
      <scale value="0.1"/> <rotate angle="30" z="1"/> </diffuse>
    <specular value="0.2"/> <glossiness value="50"/> <reflection value="0.3"/>
  </material>
  <material type="blinn" name="glass"> <diffuse value="0.05"/> <specular value="0.6"/> <glossiness value="40"/>
    <refraction value="0.85" index="1.4"/> <absorption r="0.05" g="0.01" b="0.02"/> </material>
  <material type="blinn" name="mir"> <diffuse value="0.1"/> <specular value="0.8"/> <glossiness value="8"/> <reflection value="0.8"/> </material>
  <material type="microfacet" name="wht"/>
  <light type="direct"> <intensity value="0.6"/> <direction x="1" y="1" z="-2"/> </light>
  <light type="point"> <intensity value="60" r="1" g="0.9" b="0.8"/> <position x="0" y="-2" z="9"/> <size value="1.5"/> <attenuation value="1"/> </light>
  <light type="ambient"> <intensity value="0.2"/> </light>
  <background r="0.2" g="0.3" b="0.5"/>
  <environment value="0.7" texture="checkerboard"> <color1 r="0.3" g="0.5" b="0.9"/> <color2 r="0.9" g="0.9" b="0.9"/> <scale value="0.05"/> </environment>
</scene>
<camera gamma="sRGB"> <position x="0" y="-18" z="6"/> <target x="0" y="0" z="2"/> <up x="0" y="0" z="1"/> <fov value="40"/>
  <focaldist value="18"/> <dof value="0.15"/> <width value="96"/> <height value="64"/> </camera></xml>"""

EMPTY_SCENE = """<xml><scene><background r="0.25" g="0.5" b="0.75"/>
<light type="point"><intensity value="10"/><position x="0" y="0" z="5"/></light></scene>
<camera><position x="0" y="-10" z="0"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="40"/>
<width value="40"/><height value="30"/></camera></xml>"""

UNLIT_SCENE = """<xml><scene>
<object type="sphere" name="a" material="m"><scale value="2"/></object>
<object type="plane" name="p"><scale value="30"/><translate z="-2"/></object>
<material type="blinn" name="m"><diffuse value="0.8"/><emission r="0.1" g="0.2" b="0.3"/></material>
</scene><camera><position x="0" y="-10" z="1"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="40"/>
<width value="40"/><height value="30"/></camera></xml>"""


def check_boundary_extras(r):
    """rt_alloc_pinned / rt_free_pinned and rt_context_create_multi: a frame rendered into page-locked images equals
    the frame rendered into plain arrays; a context made by the multi-device call renders the same frame; duplicate or bad
    device ids fail and leave nothing behind."""
    import ctypes as C
    from conftest import scene_path
    from raytracer_amd import Renderer, Scene
    lib = r._lib
    scene = Scene.load(scene_path("cornell_box.xml"))
    r.LoadScene(scene)
    p = params(96, 64, spp=2, jitter=1, bounces=2)
    a = r.Render(p)
    b = r.Render(p, pinned=True)
    assert len(b._pinned) == 8
    for name in ("img", "sample_count", "hit_node", "hit_face"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.allclose(a.radiance, b.radiance, rtol=1e-5, atol=1e-6) and np.allclose(a.zbuffer, b.zbuffer)
    kept = b.radiance
    b.release()
    assert b._pinned == [] and np.array_equal(np.asarray(b.radiance), np.asarray(b.radiance)) and kept is not b.radiance
    ptr = C.c_void_p()
    assert lib.rt_alloc_pinned(0, C.byref(ptr)) != 0 and not ptr.value
    assert lib.rt_alloc_pinned(4096, None) != 0
    lib.rt_free_pinned(None)
    (r2,) = Renderer.create_multi([0], _lib=lib)
    r2.LoadScene(scene)
    c = r2.Render(p)
    assert np.array_equal(a.hit_node, c.hit_node) and np.allclose(a.radiance, c.radiance, rtol=1e-5, atol=1e-6)
    r2.close()
    out = (C.c_void_p * 2)()
    ids = (C.c_int * 2)(0, 0)
    assert lib.rt_context_create_multi(2, ids, out) != 0 and not out[0] and not out[1]        # one context per GPU
    ids = (C.c_int * 2)(0, 1 << 20)
    assert lib.rt_context_create_multi(2, ids, out) != 0 and not out[0] and not out[1]        # all or nothing
    assert lib.rt_context_create_multi(0, None, out) != 0 and lib.rt_context_create_multi(1, None, None) != 0


def check_extended_materials(r, w=160, h=90, spp=4, photons=True):
    """SURVEY 8f n4, RT_EXT_MATERIALS: this build's own MtlPhong / MtlMicrofacet (the reference's are stubs that return white,
    materials.cpp:392-408) against their restatement in the oracle.  Without the extension both sides render the reference's
    white; with it: deterministic frame ids exact and colour within 1e-4 relative, stochastic frame (reflection + transmission
    children, soft shadows) within the stochastic gate, photon deposits on microfacet surfaces gathered alike."""
    import oracle_ffi
    from conftest import scene_path
    from raytracer_amd import Scene
    from raytracer_amd._ffi import RT_EXT_MATERIALS, RT_MTL_MICROFACET, RT_MTL_PHONG
    scene = Scene.load(scene_path("ext_materials.xml"))
    kinds = [scene.desc.materials[i].kind for i in range(scene.desc.n_materials)]
    assert kinds.count(RT_MTL_MICROFACET) == 4 and kinds.count(RT_MTL_PHONG) == 1
    r.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    p0 = params(w, h, spp=1, jitter=0, bounces=0)
    g0, c0 = r.Render(p0), orc.render(p0)
    assert np.array_equal(g0.hit_node, c0.hit_node) and rel_colour_diff(g0.radiance, c0.radiance) <= 1e-4
    on_stub = np.isin(c0.hit_node, [i for i in range(scene.desc.n_nodes) if scene.node_name(i) in ("phong", "plastic", "gold", "glass", "floor")])
    assert on_stub.any() and np.allclose(c0.radiance[on_stub][:, :3], 1.0), "the reference's stubs shade white"
    p1 = params(w, h, spp=1, jitter=0, bounces=0)
    p1.extensions = RT_EXT_MATERIALS
    g1, c1 = r.Render(p1), orc.render(p1)
    assert np.array_equal(g1.hit_node, c1.hit_node)
    assert rel_colour_diff(g1.radiance, c1.radiance) <= 1e-4, rel_colour_diff(g1.radiance, c1.radiance)
    assert np.abs(c1.radiance[on_stub][:, :3] - 1.0).mean() > 0.2, "the models replace the white stubs"
    for name in ("phong", "plastic", "gold", "floor"):   # every model is lit and not constant over its object
        sel = c1.hit_node == [i for i in range(scene.desc.n_nodes) if scene.node_name(i) == name][0]
        assert sel.sum() > 50 and c1.radiance[sel][:, :3].std() > 1e-3 and c1.radiance[sel][:, :3].max() > 0.02, name
    p2 = params(w, h, spp=spp, jitter=1, bounces=3)
    p2.extensions = RT_EXT_MATERIALS
    g2, (c2, cnt) = r.Render(p2), orc.render(p2, want_counters=True)
    assert cnt["rays_secondary"] > 0.3 * cnt["rays_primary"]
    print("extended materials", assert_stochastic_close(g2, c2, "ext materials", frac=0.999, psnr=70.0, mean=1e-5))
    bad = params(w, h)
    bad.extensions = 1 << 7
    try:
        r.Render(bad)
        assert False, "unknown extension bits must be refused"
    except RuntimeError:
        pass
    if photons:
        import photon_checks
        q = photon_checks.photon_params(3000, w=64, h=36, spp=2)
        q.extensions = RT_EXT_MATERIALS
        r.LoadScene(scene)
        g3 = r.Render(q)
        n_glob, n_caus = len(r.GetPhotonMap()), len(r.GetCausticsMap())
        assert n_glob == 3000 and n_caus > 100, (n_glob, n_caus)   # microfacet surfaces take deposits, specular / transmitted paths exist
        for which in (0, 1):
            orc.photons_set_tree(which, r.GetPhotonTree(which))
        c3 = orc.render(q)
        mean = float(np.abs(g3.radiance[..., :3] - c3.radiance[..., :3]).mean())
        assert psnr8(g3.img, c3.img) >= 60.0 and mean <= 1e-4, (psnr8(g3.img, c3.img), mean)
        q0 = photon_checks.photon_params(3000, w=64, h=36, spp=2)
        r.Render(q0)     # without the extension the stubs absorb every photon that reaches them (a different map)
        assert len(r.GetCausticsMap()) < n_caus, "the stubs' GenerateSample returns false"


TEXFILTER = """<xml><scene>
  <object type="plane" name="floor" material="fine"> <scale value="60"/> </object>
  <object type="sphere" name="globe" material="clouds"> <scale value="3"/> <rotate angle="25" x="1"/> <translate x="-5" y="2" z="3"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_mapks.obj"> <scale value="2.2"/> <rotate angle="-30" z="1"/> <translate x="5" y="1" z="0"/> </object>
  <object type="plane" name="wall" material="bricks"> <scale value="12"/> <rotate angle="90" x="1"/> <translate y="30" z="12"/> </object>
  <material type="blinn" name="fine">
    <diffuse texture="checkerboard"> <color1 r="0.95" g="0.95" b="0.9"/> <color2 r="0.05" g="0.05" b="0.1"/> <scale value="0.01"/> </diffuse>
    <specular value="0"/> </material>
  <material type="blinn" name="clouds"> <diffuse value="0.9" texture="scenes/tex_clouds.png"> <scale x="0.25" y="0.5" z="1"/> </diffuse>
    <specular value="0.3"/> <glossiness texture="scenes/tex_bricks.png" value="60"/> </material>
  <material type="blinn" name="bricks"> <diffuse texture="scenes/tex_bricks.png"> <scale value="0.2"/> </diffuse> <specular value="0"/> </material>
  <light type="direct"> <intensity value="0.9"/> <direction x="0.3" y="0.8" z="-1"/> </light>
  <light type="point"> <intensity value="120"/> <position x="0" y="-6" z="12"/> <attenuation value="1"/> </light>
  <background r="0.2" g="0.3" b="0.5"/>
</scene>
<camera> <position x="0" y="-20" z="4"/> <target x="0" y="10" z="2"/> <up x="0" y="0" z="1"/> <fov value="45"/>
  <width value="160"/> <height value="90"/> </camera></xml>"""


def check_texture_filter(r, w=160, h=90, ref_spp=64):
    """SURVEY 8f n4, RT_EXT_TEX_FILTER: camera rays carry differentials into HitInfo::duvw (this build's definition) and the
    material parameters go through the reference's filtered lookup Texture::Eval(uvw, duvw) (scene.h:289-303), restated in the
    oracle.  Plane, sphere and mesh hits, checker and image textures, a textured glossiness.  Without texture maps, or with
    the extension off, nothing changes; with it the device agrees with the oracle and the 1-spp frame is nearer to a
    supersampled one than the unfiltered 1-spp frame is."""
    import oracle_ffi
    from conftest import ROOT
    from raytracer_amd import Scene
    from raytracer_amd._ffi import RT_EXT_TEX_FILTER
    scene = Scene.from_string(TEXFILTER, asset_dir=ROOT)
    assert scene.warnings == 0
    r.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    p0 = params(w, h, spp=1, jitter=0, bounces=0)
    p1 = params(w, h, spp=1, jitter=0, bounces=0)
    p1.extensions = RT_EXT_TEX_FILTER
    g0, g1, c0, c1 = r.Render(p0), r.Render(p1), orc.render(p0), orc.render(p1)
    assert np.array_equal(g1.hit_node, c1.hit_node) and np.array_equal(g1.hit_face, c1.hit_face)
    assert len(np.unique(c1.hit_node[c1.hit_node >= 0])) >= 4 and (c1.hit_face >= 0).sum() > 100
    assert rel_colour_diff(g0.radiance, c0.radiance) <= 1e-4
    d = np.abs(g1.radiance[..., :3] - c1.radiance[..., :3]) / np.maximum(np.abs(c1.radiance[..., :3]), 1e-3)
    within = (d.max(axis=-1) <= 1e-4).mean()
    mean = float(np.abs(g1.radiance[..., :3] - c1.radiance[..., :3]).mean())
    print(f"texture filter: {within:.5f} of the pixels within 1e-4, PSNR {psnr8(g1.img, c1.img):.1f} dB, mean |diff| {mean:.3g}")
    # (a tap that lands within an ulp of a checker edge or a texel border may fall on the other side: 1/32 of one pixel's colour)
    assert within >= 0.999 and psnr8(g1.img, c1.img) >= 60.0 and mean <= 1e-5, (within, psnr8(g1.img, c1.img), mean)
    assert np.allclose(g1.albedo, c1.albedo, atol=2e-3)
    changed = np.abs(c1.radiance[..., :3] - c0.radiance[..., :3]).max(axis=-1) > 1e-3
    assert changed.mean() > 0.1, "the filter changes nothing"
    # against a supersampled frame (jittered, unfiltered, oracle): the filtered 1-spp frame is the nearer one on every object kind
    pr = params(w, h, spp=ref_spp, jitter=1, bounces=0)
    ref = orc.render(pr, 8)
    for name in ("floor", "globe", "wall"):
        node = [i for i in range(scene.desc.n_nodes) if scene.node_name(i) == name][0]
        sel = (c1.hit_node == node)
        e0 = float(((c0.radiance[sel][:, :3] - ref.radiance[sel][:, :3]) ** 2).mean())
        e1 = float(((c1.radiance[sel][:, :3] - ref.radiance[sel][:, :3]) ** 2).mean())
        print(f"  {name}: mse unfiltered {e0:.3g}, filtered {e1:.3g}")
        assert sel.sum() > 100 and e1 < e0, (name, e0, e1)
    # secondary hits carry no derivatives and a scene without texture maps none at all: bit-equal frames
    plain = Scene.load(__import__("conftest").scene_path("spheres_room.xml"))
    r.LoadScene(plain)
    q0 = params(96, 64, spp=2, jitter=1, bounces=2)
    q1 = params(96, 64, spp=2, jitter=1, bounces=2)
    q1.extensions = RT_EXT_TEX_FILTER
    a, b = r.Render(q0), r.Render(q1)
    assert np.array_equal(a.hit_node, b.hit_node) and np.allclose(a.radiance, b.radiance, rtol=1e-6, atol=1e-7)
    # a camera that skims the floor: the neighbouring pixels' rays meet the plane arbitrarily far away (derivatives held to
    # RT_DUVW_MAX), some not at all; the frame still agrees with the oracle
    low = Scene.from_string(TEXFILTER.replace('<position x="0" y="-20" z="4"/> <target x="0" y="10" z="2"/>',
                                              '<position x="0" y="-20" z="0.02"/> <target x="0" y="10" z="0.02"/>'), asset_dir=ROOT)
    r.LoadScene(low)
    gl, cl = r.Render(p1), oracle_ffi.OracleScene(low).render(p1)
    assert np.array_equal(gl.hit_node, cl.hit_node) and np.isfinite(gl.radiance).all() and np.isfinite(cl.radiance).all()
    dl = np.abs(gl.radiance[..., :3] - cl.radiance[..., :3]) / np.maximum(np.abs(cl.radiance[..., :3]), 1e-3)
    assert (dl.max(axis=-1) <= 1e-4).mean() >= 0.995 and psnr8(gl.img, cl.img) >= 50.0, ((dl.max(axis=-1) <= 1e-4).mean(), psnr8(gl.img, cl.img))
    # stochastic frame with bounces on the textured scene: camera hits filtered, secondary hits not
    r.LoadScene(scene)
    s1 = params(w, h, spp=2, jitter=1, bounces=2)
    s1.extensions = RT_EXT_TEX_FILTER
    print("texture filter, stochastic", assert_stochastic_close(r.Render(s1), orc.render(s1), "texture filter", frac=0.998, psnr=60.0, mean=1e-5))


def check_edge_cases(r):
    """Empty scene, scene without lights, a default-material node, ragged image sizes (partial 32x32 tiles,
    partial 64-path wave tiles), 1x1 image, empty ray batches."""
    import oracle_ffi
    from raytracer_amd import Scene
    empty = Scene.from_string(EMPTY_SCENE)
    r.LoadScene(empty)
    for w, h in ((40, 30), (1, 1), (33, 17), (65, 3)):
        p = params(w, h, spp=2, jitter=1, bounces=2)
        g = r.Render(p)
        assert (g.hit_node == -1).all() and (g.sample_count == 2).all()
        assert np.allclose(g.radiance[..., :3], [0.25, 0.5, 0.75], atol=1e-6)
        assert (g.zbuffer > 1e30).all() and np.allclose(g.albedo, [0.25, 0.5, 0.75], atol=1e-6)
    assert len(r.TraceRay(random_rays(0, 1))) == 0
    assert len(r.TraceShadowRay(random_rays(0, 1), np.zeros(0, np.float32))) == 0
    hits = r.TraceRay(random_rays(100, 2))
    assert (hits["node"] == -1).all() and (hits["light"] == -1).all()
    unlit = Scene.from_string(UNLIT_SCENE)
    r.LoadScene(unlit)
    orc = oracle_ffi.OracleScene(unlit)
    for w, h in ((40, 30), (33, 17), (7, 129)):
        p = params(w, h, spp=3, jitter=1, bounces=2)
        g, c = r.Render(p), orc.render(p)
        assert np.array_equal(g.hit_node, c.hit_node)
        assert np.allclose(g.radiance, c.radiance, rtol=1e-5, atol=1e-6)
        on_sphere = c.hit_node == 1
        interior = np.isclose(c.radiance[on_sphere][:, :3], [0.1, 0.2, 0.3], atol=1e-6).all(axis=1)           # emission only
        assert on_sphere.any() and interior.mean() > 0.6   # (silhouette pixels: some samples miss the sphere)
        dark = (np.abs(c.radiance[c.hit_node == 2][:, :3]) <= 1e-7).all(axis=1)                                # default Blinn, no light
        assert dark.mean() > 0.9


def random_scene_xml(seed, n_objects=14, width=96, height=64, light_size=None):
    """A seeded random scene graph: spheres, planes and small procedural meshes under nested nodes (up to 4 levels) with
    non-uniform scales, rotations about arbitrary axes and translations in document order (xmlload.cpp:176-198),
    overlapping on purpose (equal-t ties, objects inside objects), mixed materials, one point and one direct light."""
    rng = np.random.default_rng(seed)
    kinds = ['sphere', 'sphere', 'plane', 'obj']

    def xform(scale_lo, scale_hi, spread):
        parts = []
        for _ in range(int(rng.integers(1, 4))):
            c = int(rng.integers(0, 3))
            if c == 0:
                if rng.random() < 0.5:
                    parts.append(f'<scale value="{rng.uniform(scale_lo, scale_hi):.4f}"/>')
                else:
                    parts.append(f'<scale x="{rng.uniform(scale_lo, scale_hi):.4f}" y="{rng.uniform(scale_lo, scale_hi):.4f}" z="{rng.uniform(scale_lo, scale_hi):.4f}"/>')
            elif c == 1:
                a = rng.normal(size=3)
                parts.append(f'<rotate angle="{rng.uniform(-180, 180):.3f}" x="{a[0]:.4f}" y="{a[1]:.4f}" z="{a[2]:.4f}"/>')
            else:
                t = rng.uniform(-spread, spread, size=3)
                parts.append(f'<translate x="{t[0]:.4f}" y="{t[1]:.4f}" z="{t[2]:.4f}"/>')
        return "".join(parts)

    def obj(i, depth):
        k = kinds[int(rng.integers(0, len(kinds)))]
        name = f'gen:icosphere?seed={i}&amp;tris=320' if k == 'obj' else f'o{i}'
        children = ""
        if depth < 3 and rng.random() < 0.45:
            children = "".join(obj(i * 7 + j + 100, depth + 1) for j in range(int(rng.integers(1, 3))))
        mat = ['m0', 'm1', 'm2', 'm3'][int(rng.integers(0, 4))]
        return f'<object type="{k}" name="{name}" material="{mat}">{xform(0.6, 3.0, 5.0)}{children}</object>'

    body = "".join(obj(i, 0) for i in range(n_objects))
    size = "" if light_size is None else f'<size value="{light_size}"/>'   # (a sized point light: soft shadows)
    return f"""<xml><scene>{body}
<object name="grp">{xform(0.8, 1.5, 3.0)}<object type="plane" name="floor" material="m0"><scale value="25"/><translate z="-6"/></object></object>
<material type="blinn" name="m0"><diffuse r="0.7" g="0.6" b="0.5"/><specular value="0.3"/><glossiness value="30"/></material>
<material type="blinn" name="m1"><diffuse value="0.1"/><specular value="0.8"/><glossiness value="60"/><reflection value="0.7"/></material>
<material type="blinn" name="m2"><diffuse value="0.05"/><specular value="0.5"/><glossiness value="80"/><refraction value="0.9" index="1.45"/><absorption r="0.02" g="0.05" b="0.01"/></material>
<material type="phong" name="m3"/>
<light type="point"><intensity value="80"/><position x="3" y="-6" z="12"/>{size}<attenuation value="1"/></light>
<light type="direct"><intensity value="0.4"/><direction x="-1" y="2" z="-3"/></light>
<background r="0.1" g="0.1" b="0.2"/>
</scene><camera><position x="0" y="-26" z="6"/><target x="0" y="0" z="0"/><up x="0" y="0" z="1"/><fov value="42"/>
<width value="{width}"/><height value="{height}"/></camera></xml>"""


def check_random_scene_graphs(r, seeds, n_rays, w, h):
    """Bit-exact decisions on randomly nested, transformed, overlapping scene graphs: batched closest / any-hit queries
    and a deterministic frame, with the linear instance walk and with the instance BVH."""
    import os
    import oracle_ffi
    from raytracer_amd import RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK, Scene
    for seed in seeds:
        xml = random_scene_xml(seed, width=w, height=h)
        for tlas_min in (1000000, 2):
            r.set_option("tlas_min", tlas_min)
            try:
                scene = Scene.from_string(xml)
                r.LoadScene(scene)
            finally:
                r.set_option("tlas_min", 24)
            orc = oracle_ffi.OracleScene(scene)
            rays = random_rays(n_rays, seed=100 + seed, origin_box=((-20, -30, -8), (20, 20, 16)), target_box=((-8, -8, -6), (8, 8, 8)))
            for side in (RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK):
                assert_hits_equal(r.TraceRay(rays, side), orc.trace_closest(rays, side), f"seed {seed} tlas_min {tlas_min} side {side}")
            tmax = np.full(len(rays), 1.2, np.float32)
            tmax[::5] = np.finfo(np.float32).max
            assert np.array_equal(r.TraceShadowRay(rays, tmax), orc.trace_any(rays, tmax)), f"seed {seed}: occlusion differs"
            p = params(w, h, spp=1, jitter=0, bounces=0)
            g, c = r.Render(p), orc.render(p)
            assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face), f"seed {seed}: hit ids differ"
            assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4, f"seed {seed}"
            p = params(w // 2, h // 2, spp=2, jitter=1, bounces=3)
            g, c = r.Render(p), orc.render(p)
            assert_stochastic_close(g, c, f"random scene graph, seed {seed}", frac=0.999)


# ---------------------------------------------------------------------------------------------- stochastic frames
def stochastic_report(g, c, rel=1e-3):
    """How close two renders at the same Philox seed / spp are: the share of pixels whose radiance agrees within `rel`
    (relative to max(|c|, 1)), the largest and mean absolute difference, and the PSNR of the 8-bit images."""
    a = g.radiance[..., :3].astype(np.float64)
    b = c.radiance[..., :3].astype(np.float64)
    d = np.abs(a - b)
    ok = (d <= rel * np.maximum(np.abs(b), 1.0)).all(axis=-1)
    return {"frac": float(ok.mean()), "max": float(d.max()), "mean": float(d.mean()), "psnr": float(psnr8(g.img, c.img))}


def assert_stochastic_close(g, c, what, frac=0.9995, rel=1e-3, psnr=80.0, mean=1e-6):
    """Gate for stochastic configurations at equal seed and spp.  The device draws the same Philox numbers as the oracle;
    what differs is the last ulp of sinf / cosf / powf / expf (device library vs libm) and the order of float sums, so
    almost every pixel agrees to ~1e-6 and a few differ visibly where an ulp flips a hit / miss or a below-surface
    decision of one sample.  Required: >= `frac` of the pixels within `rel`, 8-bit PSNR >= `psnr` dB, mean |diff| <= `mean`.
    (Measured on MI355X, round 2: every pixel within 1e-3, largest difference 7e-4, mean 3e-8 .. 6e-8, PSNR 95 .. 103 dB.)"""
    s = stochastic_report(g, c, rel)
    msg = f"{what}: {s}"
    assert np.array_equal(g.sample_count, c.sample_count), msg
    assert s["frac"] >= frac, msg
    assert s["psnr"] >= psnr, msg
    assert s["mean"] <= mean, msg
    return s


def check_monte_carlo_indirect(r, scene, w, h, spp, strict=None):
    """SampleIndirectDiffuseCosin (materials.cpp:175-212, call site :309-314) with monteCarloBounces = 2,
    three specular bounces and the scene's soft shadows (16..128 rays): BASELINE config 3's integrator."""
    import oracle_ffi
    r.LoadScene(scene)
    p = params(w, h, spp=spp, jitter=1, bounces=3, mc=2)
    g, (c, cnt) = r.Render(p), oracle_ffi.OracleScene(scene).render(p, want_counters=True)
    assert cnt["rays_secondary"] > 0.5 * cnt["rays_primary"] and cnt["tex_evals"] > 0 and cnt["rays_shadow"] > 16 * cnt["rays_primary"]
    assert np.array_equal(g.hit_node, c.hit_node)
    q = params(w, h, spp=spp, jitter=1, bounces=3, mc=0)
    base = oracle_ffi.OracleScene(scene).render(q)
    assert np.abs(c.radiance[..., :3] - base.radiance[..., :3]).mean() > 1e-2, "the indirect term contributes nothing"
    if strict is not None:
        assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= strict
        return None
    return assert_stochastic_close(g, c, "MC indirect")


def check_textured_scene(r, w, h, spp, strict=None):
    """DOF lens sampling (RayTracer.cpp:152-164), environment mapping (scene.h:373-384), nested + transformed checker
    textures (texture.cpp:136-140, xmlload.cpp:370-374), a direct light, a soft point light, glossy reflection /
    refraction with absorption and a microfacet (white stub) material, in one frame."""
    import oracle_ffi
    from raytracer_amd import Scene
    scene = Scene.from_string(TEXTURED)
    assert scene.desc.n_texmaps == 3
    r.LoadScene(scene)
    p = params(w, h, spp=spp, jitter=1, bounces=3)
    g, (c, cnt) = r.Render(p), oracle_ffi.OracleScene(scene).render(p, want_counters=True)
    assert cnt["tex_evals"] > 0 and cnt["rays_secondary"] > 0
    assert np.array_equal(g.hit_node, c.hit_node)
    if strict is not None:
        assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= strict
        return None
    return assert_stochastic_close(g, c, "textured scene")


def check_adaptive_sampling(r, scene, w, h, spp_min, spp_max, threshold, bounces=2, strict=None, equal_frac=1.0):
    """RayTracer.cpp:136-204: the Student-t stop test after every sample >= spp_min (SURVEY Q1 / Q2 positions); the
    per-pixel sample counts are the observable.  `equal_frac`: share of pixels whose count must equal the oracle's (a
    pixel whose half-width sits within an ulp of the threshold may stop one sample apart on the device)."""
    import oracle_ffi
    r.LoadScene(scene)
    p = params(w, h, jitter=1, bounces=bounces)
    p.spp_min, p.spp_max = spp_min, spp_max
    p.adaptive_threshold = threshold
    g, c = r.Render(p), oracle_ffi.OracleScene(scene).render(p)
    assert c.sample_count.min() == spp_min + 1 and c.sample_count.max() == spp_max and len(np.unique(c.sample_count)) > 3
    same = g.sample_count == c.sample_count
    assert same.mean() >= equal_frac, f"sample counts equal for {same.mean():.5f} of the pixels"
    assert np.abs(g.sample_count - c.sample_count).max() <= (0 if equal_frac >= 1.0 else spp_max)
    d = np.abs(g.radiance[..., :3] - c.radiance[..., :3])
    if strict is not None:
        assert d.max() <= strict
    else:
        assert d[same].mean() <= 2e-5 and (d[same] <= 1e-3).mean() >= 0.999
    return float(same.mean())


def check_async_render(r, scene, w, h, spp):
    """BeginRender ("returns immediately", renderer.h:229) + polling GetNumRenderedPixels / IsRenderDone as the viewport
    does (viewport.cpp:583-586) + StopRender (renderer.h:230) gives the frame Render gives."""
    import time
    r.LoadScene(scene)
    p = params(w, h, spp=spp, jitter=1, bounces=2)
    ref = r.Render(p)
    r.BeginRender(p)
    seen = [r.GetNumRenderedPixels()]
    t0 = time.time()
    while not r.IsRenderDone() and time.time() - t0 < 60.0:
        seen.append(r.GetNumRenderedPixels())
        time.sleep(0.0005)
    assert r.IsRenderDone()
    img = r.StopRender()
    seen.append(r.GetNumRenderedPixels())
    assert seen[-1] == w * h and all(b >= a for a, b in zip(seen, seen[1:])) and min(seen) >= 0
    assert np.array_equal(img.hit_node, ref.hit_node) and np.array_equal(img.sample_count, ref.sample_count)
    # (float atomics into a path's radiance slot are unordered: equal to the order of the sums)
    assert np.allclose(img.radiance, ref.radiance, rtol=1e-5, atol=1e-6)
    assert (img.img == ref.img).mean() > 0.999
    # a second BeginRender while nothing runs, then a blocking Render right behind it (joins the worker first)
    r.BeginRender(p)
    again = r.Render(p)
    assert np.allclose(again.radiance, ref.radiance, rtol=1e-5, atol=1e-6)
    return seen


def check_stress_scene(r, scene, w, h, n_rays, oracle_scene=None):
    """BASELINE config 5's workload shape (rth_scene_generate_stress: icospheres over a displaced ground in ONE mesh,
    two sphere lights): node + face ids bit-exact at the pixel centre, batched closest / any-hit parity, and a
    stochastic frame with soft shadows and three bounces."""
    import oracle_ffi
    from raytracer_amd import RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK
    r.LoadScene(scene)
    orc = oracle_ffi.OracleScene(oracle_scene if oracle_scene is not None else scene)
    p = params(w, h, spp=1, jitter=0, bounces=0)
    g, c = r.Render(p), orc.render(p)
    assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face)
    assert (c.hit_face >= 0).mean() > 0.2 and len(np.unique(c.hit_face)) > 1000
    assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4
    lo = np.array(scene.desc.meshes[0].bound_min[:], np.float32)
    hi = np.array(scene.desc.meshes[0].bound_max[:], np.float32)
    pad = 0.1 * (hi - lo)
    rays = random_rays(n_rays, seed=77, origin_box=(tuple(lo - pad), tuple(hi + pad)), target_box=(tuple(lo), tuple(hi)))
    for side in (RT_HIT_FRONT, RT_HIT_FRONT_AND_BACK):
        assert_hits_equal(r.TraceRay(rays, side), orc.trace_closest(rays, side), f"stress scene side {side}")
    tmax = np.full(len(rays), 0.6, np.float32)
    tmax[::3] = np.finfo(np.float32).max
    occ = orc.trace_any(rays, tmax)
    assert np.array_equal(r.TraceShadowRay(rays, tmax), occ) and 0.05 < occ.mean() < 0.95
    return g, c


def check_shadow_hints(r, scene, w, h, spp=1, tol=5e-5, device_bvh=False):
    """Occluder hints of the soft-shadow mesh walks (k_shadow_mesh<.., HINT>): an any-hit walk may visit children in any order,
    so the frame with hints forced on equals the frame with hints off — per pixel to the order of the float atomics — and
    both equal the oracle's.  Run on the stress scene's shape (rays start on the mesh that shadows them)."""
    import oracle_ffi
    frames = []
    for on in (0, 1, 2):   # off, on (walks from the root, steered), on + walks that start around their hint (forced for this small tree)
        r.set_option("shadow_hints", min(on, 1))
        r.set_option("restart_min_depth", 0 if on == 2 else 99)
        try:
            r.LoadScene(scene)
            p = params(w, h, spp=spp, jitter=1, bounces=1)
            r.counters(enable=True, reset=True)
            frames.append(r.Render(p))
            frames[-1].node_visits = r.counters()["node_visits"]
            r.counters(enable=False)
        finally:
            r.set_option("shadow_hints", -1)
            r.set_option("restart_min_depth", 19)
    a, b, b2 = frames
    for f in (b, b2):
        assert np.array_equal(a.hit_node, f.hit_node)
        assert np.abs(a.radiance[..., :3] - f.radiance[..., :3]).max() <= 1e-5, np.abs(a.radiance[..., :3] - f.radiance[..., :3]).max()
    c = oracle_ffi.OracleScene(scene.with_bvh({0: r.GetMeshBVH(0)}) if device_bvh else scene).render(params(w, h, spp=spp, jitter=1, bounces=1))
    assert np.abs(b.radiance[..., :3] - c.radiance[..., :3]).max() <= tol, np.abs(b.radiance[..., :3] - c.radiance[..., :3]).max()
    shadowed = (c.radiance[..., :3].sum(axis=-1) < 0.5 * np.median(c.radiance[..., :3].sum(axis=-1))).mean()
    print(f"shadow hints: node visits {a.node_visits} -> {b.node_visits} ({b.node_visits / a.node_visits:.3f}), around the hint first {b2.node_visits}, dark pixels {shadowed:.2f}")
    return a.node_visits, b.node_visits


def mesh_arrays(m):
    """positions (nv, 3) and position index triples (nf, 3) of an rt_mesh"""
    pos = np.ctypeslib.as_array(m.positions, shape=(m.n_positions, 3)).copy()
    idx = np.ctypeslib.as_array(m.pos_idx, shape=(m.n_faces, 3)).copy()
    return pos, idx


def assert_bvh_valid(nodes, elems, m, max_depth=32):
    """A tree in rt_mesh form is a BVH of mesh m: every face in exactly one leaf, leaves of 1..4 faces, every child box
    inside its parent's, every triangle inside its leaf's box, no leaf deeper than the walk's stack."""
    from raytracer_amd import _ffi
    leaf = np.uint32(_ffi.RT_BVH_LEAF_BIT) if hasattr(_ffi, "RT_BVH_LEAF_BIT") else np.uint32(0x80000000)
    pos, idx = mesh_arrays(m)
    tri_lo, tri_hi = pos[idx].min(axis=1), pos[idx].max(axis=1)
    is_leaf = (nodes["b"] & leaf) != 0
    cnt = (nodes["b"][is_leaf] & ~leaf).astype(np.int64)
    assert m.n_faces <= 4 or ((cnt >= 1) & (cnt <= 4)).all()
    assert cnt.sum() == len(elems) == m.n_faces and np.array_equal(np.sort(elems), np.arange(m.n_faces, dtype=np.uint32))
    inner = np.nonzero(~is_leaf)[0]
    for side in ("a", "b"):
        ch = nodes[side][inner]
        assert (ch > inner).all() and (ch < len(nodes)).all()          # pre-order: children behind their parent
        assert (nodes["bmin"][ch] >= nodes["bmin"][inner]).all() and (nodes["bmax"][ch] <= nodes["bmax"][inner]).all()
    assert len(np.unique(np.concatenate([nodes["a"][inner], nodes["b"][inner]]))) == len(nodes) - 1   # a tree: every node but the root has one parent
    lf = np.nonzero(is_leaf)[0]
    owner = np.repeat(lf, cnt)
    first = nodes["a"][lf].astype(np.int64)
    assert np.array_equal(first, np.concatenate([[0], np.cumsum(cnt)[:-1]]))   # leaves own consecutive runs of the element list
    assert (tri_lo[elems] >= nodes["bmin"][owner]).all() and (tri_hi[elems] <= nodes["bmax"][owner]).all()
    depth = np.zeros(len(nodes), np.int32)
    for i in inner:            # pre-order: a parent is numbered before its children
        depth[nodes["a"][i]] = depth[nodes["b"][i]] = depth[i] + 1
    assert depth.max() <= max_depth
    return int(depth.max())


def check_device_bvh(r, scene, w, h, n_rays):
    """n2: TriObj::Load's BVH (objects.h:60-67).  (1) A tree that comes with the mesh survives rt_scene_commit: what
    rt_scene_mesh_bvh hands back is the tree that went in.  (2) A mesh WITHOUT a tree gets one built on the device: it is a
    valid BVH of the mesh, and with the oracle walking that same tree the hit ids are bit-exact and the batched queries agree
    (check_stress_scene)."""
    r.LoadScene(scene)
    nm = scene.desc.n_meshes
    for i in range(nm):
        m = scene.desc.meshes[i]
        nodes, elems = r.GetMeshBVH(i)
        ref = np.ctypeslib.as_array(C.cast(m.bvh, C.POINTER(C.c_uint32)), shape=(m.n_bvh_nodes * 8,))
        assert np.array_equal(nodes.view(np.uint32), ref), "host-built tree changed on the way through the commit"
        assert np.array_equal(elems, np.ctypeslib.as_array(m.bvh_elements, shape=(m.n_bvh_elements,)))
    bare = scene.with_bvh({i: None for i in range(nm)})
    r.LoadScene(bare)
    trees = {i: r.GetMeshBVH(i) for i in range(nm)}
    depths = [assert_bvh_valid(trees[i][0], trees[i][1], scene.desc.meshes[i]) for i in range(nm)]
    g, c = check_stress_scene(r, bare, w, h, n_rays, oracle_scene=scene.with_bvh(trees))
    return trees, depths


def check_soft_shadow_clear_view(r, scene, w, h, spp=1, tol=2e-6, min_saved=0.2, strict=None, gate=None, light_scale=1.0, shadow=None,
                                 with_oracle=True, bounces=1, mc=0):
    """Soft-shadow groups with a proven clear view of their whole light, or proven to lie in a sphere's full shadow, are
    finished without a ray (cone_clear): the frame must be the one that traces every group (same groups, same terms; only
    the order of the float atomics on a path's radiance differs), with the oracle as the third opinion, and the shortcut
    must actually save rays."""
    import oracle_ffi
    r.LoadScene(scene)
    p = params(w, h, spp=spp, jitter=1, bounces=bounces, light_scale=light_scale, mc=mc)
    if shadow is not None:   # (the reference's constants are 16 / 128, raytracer.h:17-23)
        p.shadow_min, p.shadow_max = shadow
    out = {}
    for mode in (1, 0):
        r.set_option("soft_clear", mode)
        r.counters(enable=True, reset=True)
        g = r.Render(p)
        cnt = r.counters()
        out[mode] = (g, cnt["rays_shadow"], cnt["rays_shadow_proven"])
    r.counters(enable=False)
    r.set_option("soft_clear", 1)
    (g1, n1, proven), (g0, n0, none) = out[1], out[0]
    # a proven group stands for exactly the rays it would have traced: shadow_min (clear view) or shadow_max (full shadow)
    assert none == 0 and n1 + proven == n0, f"traced {n1} + proven {proven} != {n0}"
    assert np.array_equal(g1.hit_node, g0.hit_node)
    assert rel_colour_diff(g1.radiance, g0.radiance, floor=1e-2) <= tol, "the clear-view shortcut changed the frame"
    assert n1 <= (1.0 - min_saved) * n0, f"no rays saved: {n1} vs {n0}"
    if not with_oracle:   # (frames the oracle would take minutes for: the traced frame is the reference here)
        return n1, n0
    c = oracle_ffi.OracleScene(scene).render(p)
    if strict is not None:
        assert rel_colour_diff(g1.radiance, c.radiance, floor=1e-2) <= strict
    else:
        assert_stochastic_close(g1, c, "clear-view soft shadows", **(gate or {}))
    return n1, n0


def check_soft_shadow_clear_view_random(r, seeds, w, h, strict=None, tol=2e-6):
    """The clear-view / umbra proofs on randomly nested, non-uniformly scaled, rotated, overlapping spheres, planes and
    meshes (shade points inside other objects, lights inside objects) with lights of three sizes.
    The strict comparison is with / without the proofs on the same device; against the oracle a device frame of these
    scenes has the odd pixel where an ulp of sinf / cosf flips one sample's hit (measured: 1 of 19 200 pixels), which a
    small frame's PSNR does not absorb — gate: 99.9 % of the pixels within 1e-3, as check_random_scene_graphs."""
    gate = dict(frac=0.999, psnr=35.0, mean=1e-3)
    from raytracer_amd import Scene
    try:
        tot1 = tot0 = 0
        for seed in seeds:
            r.set_option("tlas_min", 1000000 if seed % 2 else 2)   # (rays through the instance table / the instance BVH)
            # with the instance BVH, every other scene's proofs walk that BVH too (cone_clear_tlas) instead of the table
            r.set_option("cone_max_inst", 0 if seed % 4 == 0 else 64)
            xml = random_scene_xml(seed, width=w, height=h, light_size=(0.5, 1.5, 3.0)[seed % 3])
            n1, n0 = check_soft_shadow_clear_view(r, Scene.from_string(xml), w, h, strict=strict, tol=tol, min_saved=-1.0, gate=gate)
            tot1, tot0 = tot1 + n1, tot0 + n0
    finally:
        r.set_option("tlas_min", 24)
        r.set_option("cone_max_inst", 256)
    assert tot1 < 0.9 * tot0
    return tot1, tot0


def check_soft_shadow_proofs_many_instances(r, w, h, strict=None, tol=2e-6):
    """Scenes of more than 64 instances: the proofs walk the instance BVH with the group's cone (cone_clear_tlas).  The
    reference's own showcase scene (main.cpp:12: finalProject.xml, 70 instances, two lights of size 4 and 8; fixture with a
    procedural teapot) and a Cornell box walked both ways (table / BVH): the frame with the proofs is the traced frame, rays
    are saved, and the two walks prove the same groups."""
    import os
    from conftest import GOLDEN, ROOT, scene_path
    from raytracer_amd import Scene
    fp = Scene.load(os.path.join(GOLDEN, "scenes", "ref_finalProject.xml"), asset_dir=ROOT)
    assert fp.desc.n_nodes > 70
    gate = dict(frac=0.999, psnr=60.0, mean=1e-5)
    out = {}
    try:
        # 70 instances: by default the proofs and the shadow rays walk the instance table (64-bit masks of the instances a group's
        # rays still have to be tested against); forced: both through the instance BVH
        for name, cone_max, shadow_min in (("table", 256, 256), ("bvh", 0, 2)):
            r.set_option("cone_max_inst", cone_max)
            r.set_option("shadow_tlas_min", shadow_min)
            out["finalProject " + name] = check_soft_shadow_clear_view(r, fp, w, h, strict=strict, tol=tol, min_saved=0.3, gate=gate, bounces=2)
        a, b = out["finalProject table"], out["finalProject bvh"]
        assert a[1] == b[1] and a[0] <= b[0] <= 1.1 * a[0], out
        r.set_option("shadow_tlas_min", 256)
        cb = Scene.load(scene_path("cornell_box.xml"))
        r.set_option("tlas_min", 2)
        both = {}
        for cm in (256, 0):
            r.set_option("cone_max_inst", cm)
            both[cm] = check_soft_shadow_clear_view(r, cb, w, h, strict=strict, tol=tol, gate=gate)
        # (the BVH walk gives up after a step budget and never proves more than the table walk does)
        assert both[0][1] == both[256][1] and both[256][0] <= both[0][0] <= 1.1 * both[256][0], both
        out["cornell"] = both
    finally:
        r.set_option("tlas_min", 24)
        r.set_option("cone_max_inst", 256)
        r.set_option("shadow_tlas_min", 256)
    return out


def check_reference_scene_fixtures(r, w, h, strict=None):
    """north_star: "every bundled scene renders within tolerance of the CPU reference".  tests/golden/scenes/ref_*.xml carry
    the scene data of sixteen bundled reference scenes (nested transforms, direct + ambient + soft point lights, reflection /
    refraction / absorption, nested checkerboards, image textures on objects, background and environment, depth of field,
    nodes without material or lights, 49 glass spheres) to the GPU box; the reference's own files are read in place by
    test_emu_parity where the checkout exists."""
    import glob
    import os
    import oracle_ffi
    from conftest import GOLDEN, ROOT
    from raytracer_amd import Scene
    files = sorted(glob.glob(os.path.join(GOLDEN, "scenes", "ref_*.xml")))
    assert len(files) >= 16
    for f in files:
        scene = Scene.load(f, asset_dir=ROOT)
        assert scene.warnings == 0, f
        r.LoadScene(scene)
        orc = oracle_ffi.OracleScene(scene)
        # deterministic frame (BASELINE config 1's settings: pixel centre, one shadow ray to every light's centre): ids bit-exact,
        # every pixel's colour within 1e-4
        p = params(w, h, spp=1, jitter=0, bounces=0, light_scale=0.0)
        g, c = r.Render(p), orc.render(p)
        assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face), f
        assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4, f
        # the same frame with the scene's sized lights (16..128 disc samples per group, sinf / cosf per sample): a sample that an
        # ulp of the device's sincosf moves across a silhouette changes its pixel by 1 / 16 or 1 / 128 of a light's term
        # (measured: 1 of 57 600 pixels of finalProject, its 64 glass spheres under two lights of size 4 and 8)
        p = params(w, h, spp=1, jitter=0, bounces=0)
        g, c = r.Render(p), orc.render(p)
        assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face), f
        a, b = g.radiance[..., :3].astype(np.float64), c.radiance[..., :3].astype(np.float64)
        within = ((np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max(axis=-1) <= 1e-4)
        assert within.mean() >= (1.0 if strict is not None else 0.9995), (f, within.mean())
        p = params(w // 2, h // 2, spp=2, jitter=1, bounces=2)
        p.shadow_min, p.shadow_max = 4, 8
        g, c = r.Render(p), orc.render(p)
        if strict is not None:
            assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= strict, f
        else:
            # (few shadow samples per group, 4..8: one shadow ray that an ulp flips at a silhouette changes a whole group's
            # early-out, so single pixels differ visibly — measured 4 of 14 400 pixels, 70 dB, on ref_soft_shadow_scene)
            assert_stochastic_close(g, c, os.path.basename(f), frac=0.999, psnr=60.0, mean=1e-5)


MULTIMTL = """<xml><scene>
  <object type="plane" name="floor" material="grey"> <scale value="30"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_illum2.obj"> <scale value="1.6"/> <translate x="-9" y="2" z="0"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_illum3.obj"> <scale value="1.6"/> <rotate angle="40" z="1"/> <translate x="-3" y="4" z="0"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_illum6.obj"> <scale value="1.6"/> <translate x="3" y="2" z="0"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_mapks.obj"> <scale value="1.6"/> <rotate angle="-30" z="1"/> <translate x="9" y="4" z="0"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_illum3.obj"> <scale value="0.8"/> <translate x="-1" y="-5" z="0"/> </object>
  <object type="obj" name="tests/golden/meshes/mm_illum6.obj" material="grey"> <scale value="0.8"/> <translate x="4" y="-5" z="0"/> </object>
  <material type="blinn" name="grey"> <diffuse value="0.6"/> <specular value="0.1"/> </material>
  <material type="blinn" name="tests/golden/meshes/mm_mapks.obj"> <diffuse r="0.9" g="0.1" b="0.9"/> </material>
  <light type="point"> <intensity value="300"/> <position x="-4" y="-12" z="16"/> <size value="1"/> <attenuation value="1"/> </light>
  <light type="direct"> <intensity value="0.3"/> <direction x="1" y="2" z="-2"/> </light>
  <background r="0.1" g="0.15" b="0.3"/> <environment r="0.3" g="0.4" b="0.6"/>
</scene>
<camera gamma="sRGB"> <position x="0" y="-26" z="9"/> <target x="0" y="1" z="2"/> <up x="0" y="0" z="1"/> <fov value="40"/>
  <width value="160"/> <height value="90"/> </camera></xml>"""


def load_multimtl_scene():
    """Four of this repository's own OBJ + MTL pairs (tests/golden/meshes, usemtl groups, illum 2 / 3 / 6, map_Kd, map_Ks, Ni,
    Tf) on nodes WITHOUT material=: LoadNode -> SetNodeMaterials -> CreateMultiMtl (xmlload.cpp:159, 294-343)."""
    from conftest import ROOT
    from raytracer_amd import Scene
    return Scene.from_string(MULTIMTL, asset_dir=ROOT)


def check_multimtl_scene(r, w, h, strict=None):
    """The frame of the MultiMtl scene against the oracle: ids bit-exact, colour 1e-4 at the pixel centre, the stochastic
    frame (reflection and refraction from illum 3 / 6, image textures from map_Kd / map_Ks, soft shadows) within the gate."""
    import oracle_ffi
    scene = load_multimtl_scene()
    assert scene.warnings == 0
    r.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    p = params(w, h, spp=1, jitter=0, bounces=0)
    g, c = r.Render(p), orc.render(p)
    assert np.array_equal(g.hit_node, c.hit_node) and np.array_equal(g.hit_face, c.hit_face)
    assert len(np.unique(c.hit_node[c.hit_face >= 0])) == 6, "not every mesh node is in view"
    assert rel_colour_diff(g.radiance, c.radiance) <= 1e-4
    p = params(w, h, spp=2, jitter=1, bounces=3)
    g, (c, cnt) = r.Render(p), orc.render(p, want_counters=True)
    assert cnt["tex_evals"] > 0 and cnt["rays_secondary"] > 0.05 * cnt["rays_primary"]
    if strict is not None:
        assert np.abs(g.radiance[..., :3] - c.radiance[..., :3]).max() <= strict
        return None
    return assert_stochastic_close(g, c, "MultiMtl scene", frac=0.999, psnr=60.0, mean=1e-5)


PHILOX_KAT = [  # Random123 kat_vectors, philox4x32 10 rounds: (counter, key, output)
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def check_philox_on_device(r, n=100000):
    """SURVEY a15 on the device itself: the kernels' Philox4x32-10 against the published Random123 known-answer vectors,
    against the oracle's generator on seeded random counters and keys, the float mapping (RNG::RandomFloat, rng.h:60-65:
    u * 2^-32 clamped below 1) bit for bit, and the counter layout a frame uses — pixel (x, y)'s Cranley-Patterson offsets are
    units 0 and 1 of counter (pixel, 0, 0, PIXEL tag) under the key (seed_lo, seed_hi) — through a render: with those offsets
    and the Halton points the oracle's camera ray for every sample of the pixel must be the one the device traced (the hit
    ids of a jittered 1-spp frame agree only if the device drew the same numbers)."""
    import oracle_ffi
    lib = oracle_ffi.load()
    for ctr, key, expect in PHILOX_KAT:
        words, _ = r.debug_philox(np.array([ctr], np.uint32), key[0], key[1])
        assert tuple(int(x) for x in words[0]) == expect, (ctr, key, [hex(int(x)) for x in words[0]])
    rng = np.random.default_rng(12)
    ctr = rng.integers(0, 1 << 32, size=(n, 4), dtype=np.uint64).astype(np.uint32)
    ctr[:64] = np.array([[i, 0, 0, 0] for i in range(64)], np.uint32)          # the first pixels' offset counters
    ctr[64:128, 3] = np.uint32(0xffffffff)                                     # carries in the high counter word
    for key in ((1, 0), (0xdeadbeef, 0x12345678), (0xffffffff, 0xffffffff)):
        words, units = r.debug_philox(ctr, *key)
        want = np.zeros_like(ctr)
        k = np.array(key, np.uint32)
        o = np.zeros(4, np.uint32)
        for i in list(range(256)) + list(range(256, n, max(1, n // 2000))):
            c = np.ascontiguousarray(ctr[i])
            lib.orc_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
            assert np.array_equal(words[i], o), (i, key)
        f = np.minimum(words.astype(np.float32) * np.float32(2.0 ** -32), np.float32(np.nextafter(np.float32(1.0), np.float32(0.0))))
        assert np.array_equal(units, f) and units.max() < 1.0 and units.min() >= 0.0
        assert abs(float(units.mean()) - 0.5) < 5e-3
    # (the mapping is checked on the formula above for every word of 3 x n x 4 draws, among them words above 0xff000000)
    assert (words >= np.uint32(0xff000000)).any()
    # the counter layout of a frame: units 0 / 1 of counter (pixel, 0, 0, PIXEL tag = 0) under the key (seed_lo, seed_hi) are the
    # pixel's Cranley-Patterson offsets (RayTracer.cpp:129-131 draws them from RNG(index)); with them and the Halton points the
    # oracle's camera ray of (pixel, sample) is reproduced here in numpy (RayTracer.cpp:95-101, 138-149, 165-174)
    from raytracer_amd import Scene
    from conftest import scene_path
    scene = Scene.load(scene_path("cornell_box.xml"))
    orc = oracle_ffi.OracleScene(scene)
    W, H = 64, 36
    p = params(W, H, spp=8, jitter=1, seed=77)
    p.seed_hi = 5
    cam = scene.desc.camera
    assert cam.dof == 0.0
    f32 = np.float32
    d, up, pos = np.array(cam.dir[:], f32), np.array(cam.up[:], f32), np.array(cam.pos[:], f32)
    X, Y, Z = np.cross(up, -d).astype(f32), up, -d
    wh = f32(2.0) * f32(cam.focaldist) * f32(np.tan(np.float64(f32(cam.fov) * f32(np.pi) / f32(180.0)) / 2.0))
    ww = wh * f32(W) / f32(H)
    pixels = np.array([0, 1, W + 3, 17 * W + 40, W * H - 1], np.uint32)
    ctrs = np.zeros((len(pixels), 4), np.uint32)
    ctrs[:, 0] = pixels
    _, units = r.debug_philox(ctrs, 77, 5)
    for j, pix in enumerate(pixels):
        x, y = int(pix) % W, int(pix) // W
        ro = units[j, 0]
        for i in (0, 1, 5, 7):
            hx = f32(lib.orc_halton(i, 2)) + ro
            hy = f32(lib.orc_halton(i, 3)) + ro          # (Y also uses randomOffset, RayTracer.cpp:141)
            hx = hx - f32(1) if hx > 1 else hx
            hy = hy - f32(1) if hy > 1 else hy
            px = -(ww / f32(2)) + ww * (f32(x) + f32(0.5) + hx) / f32(W)
            py = (wh / f32(2)) - wh * (f32(y) + f32(0.5) + hy) / f32(H)
            want = X * px + Y * py + Z * (-f32(cam.focaldist))
            o, got = orc.primary_ray(p, x, y, i)
            assert np.allclose(o, pos) and np.allclose(got, want, rtol=2e-6, atol=2e-6), (pix, i, got, want)
