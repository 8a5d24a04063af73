"""Denoise-filter checks shared by the emulation (CPU) and GPU tests.  The reference's Denoiser wraps Intel OIDN
(denoiser.cpp:22-64), whose weights are absent: parity against the reference is UNPINNED.  What is checked is the HIP
kernel against the CPU restatement of this build's own filter, plus filter properties."""
import numpy as np

import oracle_ffi
from common import psnr8


def synthetic_frame(w, h, seed, noise=0.08):
    """Piecewise-smooth LDR image (two discs on a gradient) + noise, with matching albedo / normal guides."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    clean = np.zeros((h, w, 3), np.float32)
    clean[..., 0] = 0.2 + 0.5 * x / w
    clean[..., 1] = 0.3 + 0.4 * y / h
    clean[..., 2] = 0.5
    albedo = np.full((h, w, 3), 0.5, np.float32)
    normal = np.zeros((h, w, 3), np.float32)
    normal[..., 2] = 1.0
    for cx, cy, rad, col in ((0.3, 0.4, 0.18, (0.9, 0.2, 0.1)), (0.7, 0.6, 0.22, (0.1, 0.8, 0.3))):
        m = (x - cx * w) ** 2 + (y - cy * h) ** 2 < (rad * min(w, h)) ** 2
        clean[m] = col
        albedo[m] = col
        normal[m] = (0.0, 0.6, 0.8)
    noisy = np.clip(clean + rng.normal(scale=noise, size=clean.shape).astype(np.float32), 0, 1).astype(np.float32)
    return clean, noisy, albedo, normal


def check_denoise(r, w, h, tol):
    clean, noisy, albedo, normal = synthetic_frame(w, h, seed=4)
    for a, n in ((None, None), (albedo, normal), (albedo, None), (None, normal)):
        g = r.Denoise(noisy, a, n)
        c = oracle_ffi.denoise(noisy, a, n)
        assert np.isfinite(g).all()
        assert np.abs(g - c).max() <= tol, f"denoise differs from the CPU restatement by {np.abs(g - c).max()}"
    # properties: a constant image is a fixed point; noise goes down and edges survive
    const = np.full((h, w, 3), 0.37, np.float32)
    assert np.abs(r.Denoise(const) - const).max() <= 1e-6
    g = r.Denoise(noisy)
    to8 = lambda im: np.clip(im * 255.0 + 0.5, 0, 255).astype(np.uint8)
    assert psnr8(to8(g), to8(clean)) >= psnr8(to8(noisy), to8(clean)) + 5.0
    gg = r.Denoise(noisy, albedo, normal)
    assert psnr8(to8(gg), to8(clean)) >= psnr8(to8(g), to8(clean))
    # odd sizes / tiny images (every tap clipped)
    for ww, hh in ((1, 1), (3, 2), (17, 5)):
        _, nz, _, _ = synthetic_frame(ww, hh, seed=1)
        assert np.abs(r.Denoise(nz) - oracle_ffi.denoise(nz)).max() <= tol
