import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_ffi, photon_checks as pc
from raytracer_amd import Renderer, Scene
scene = Scene.load('tests/golden/scenes/cornell_box.xml')
r = Renderer(0); r.LoadScene(scene)
orc = oracle_ffi.OracleScene(scene)
p = pc.photon_params(20000)
r.BuildPhotonMaps(p); orc.photons_build(p)
for which, getter in ((0, r.GetPhotonMap), (1, r.GetCausticsMap)):
    g, c = getter(), orc.photons(which)
    dpos = np.abs(g["position"] - c["position"]).max(axis=1)
    ok = dpos <= 1e-3
    rel = np.abs(g["power"] - c["power"]) / np.maximum(c["power"], 1e-12)
    print(which, len(g), 'pos match', ok.mean(), 'exact pos', (dpos == 0).mean(), 'rel power quantiles', np.quantile(rel[ok], [0.5, 0.9, 0.99, 0.999, 1.0]))
    bad = np.flatnonzero(ok & (rel > 1e-3))[:5]
    for b in bad:
        print('  ', b, g[b], c[b])
