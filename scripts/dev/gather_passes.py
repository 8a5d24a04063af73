"""Developer probe: time of the photon gather in the first (cold hints) and later passes of a C4 frame."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from raytracer_amd import Renderer

cfg = bench.CONFIGS["c4"]
r = Renderer(0)
r.LoadScene(cfg.scene())
p = cfg.params(0)
accum = torch.zeros(cfg.width * cfg.height * 8, dtype=torch.float32, device="cuda:0")
r.BuildPhotonMaps(p)
for rep in range(2):
    r.LoadScene(cfg.scene())   # a commit clears the hints
    r.BuildPhotonMaps(p)
    for i in range(4):
        p.sample_begin = 8 * i
        r.profile(enable=True, reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r.render_accumulate(p, accum.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = r.profile()
        print(f"rep {rep} pass {i}: {dt * 1e3:.1f} ms, gather {pr['photon_gather']['ms']:.1f} ms", flush=True)
