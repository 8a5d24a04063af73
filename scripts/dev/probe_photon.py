import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from raytracer_amd import Renderer, Scene, default_params
r = Renderer(0); r.LoadScene(Scene.load('tests/golden/scenes/cornell_box.xml'))
p = default_params(); p.photons = 1; p.seed_lo = 1
for i in range(4):
    p.seed_lo = 1 + i
    t = time.perf_counter(); r.BuildPhotonMaps(p); print('build ms', (time.perf_counter() - t) * 1e3)
