"""Diagnostic: where a fixture's deterministic frame differs between the device and the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi
from common import params
from raytracer_amd import Renderer, Scene
r = Renderer(0)
for name in sys.argv[1:]:
    scene = Scene.load(os.path.join(ROOT, "tests", "golden", "scenes", name), asset_dir=ROOT)
    r.LoadScene(scene)
    orc = oracle_ffi.OracleScene(scene)
    for ls in (1.0, 0.0):
        p = params(320, 180, spp=1, jitter=0, bounces=0, light_scale=ls)
        g, c = r.Render(p), orc.render(p)
        a, b = g.radiance[..., :3].astype(np.float64), c.radiance[..., :3].astype(np.float64)
        rel = (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max(axis=-1)
        bad = rel > 1e-4
        print(name, "light_scale", ls, "ids equal", np.array_equal(g.hit_node, c.hit_node), "bad px", int(bad.sum()), "max rel", rel.max())
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:6]:
            print("   px", x, y, "gpu", a[y, x], "cpu", b[y, x], "node", c.hit_node[y, x])
