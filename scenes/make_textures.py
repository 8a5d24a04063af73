#!/usr/bin/env python3
"""Generates the two image textures of the C3 scene (1024x1024 RGB PNG, seeded): a brick pattern and a cloud field.
They stand in for the reference's scenes/bricks.png and scenes/clouds.png (same size and role; the reference's
own files are not copied).  Run from the repo root: python scenes/make_textures.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bricks(n=1024, seed=3):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n]
    bh, bw, m = 64, 128, 6
    row = y // bh
    xs = x + (row % 2) * (bw // 2)
    mortar = ((y % bh) < m) | ((xs % bw) < m)
    tone = rng.random((n // bh + 1, n // bw + 2))
    t = tone[row, (xs // bw) % tone.shape[1]]
    img = np.zeros((n, n, 3), np.float32)
    img[..., 0] = 0.55 + 0.25 * t
    img[..., 1] = 0.22 + 0.12 * t
    img[..., 2] = 0.16 + 0.08 * t
    img[mortar] = (0.75, 0.73, 0.7)
    return (np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)


def clouds(n=1024, seed=5):
    rng = np.random.default_rng(seed)
    acc = np.zeros((n, n), np.float32)
    amp = 1.0
    for octave in range(2, 8):
        k = 2 ** octave
        g = rng.random((k + 1, k + 1)).astype(np.float32)
        g[-1, :] = g[0, :]
        g[:, -1] = g[:, 0]  # tiles
        u = np.linspace(0, k, n, endpoint=False)
        i = u.astype(int)
        f = (u - i).astype(np.float32)
        f = f * f * (3 - 2 * f)
        a = g[np.ix_(i, i)] * np.outer(1 - f, 1 - f) + g[np.ix_(i + 1, i)] * np.outer(f, 1 - f) \
            + g[np.ix_(i, i + 1)] * np.outer(1 - f, f) + g[np.ix_(i + 1, i + 1)] * np.outer(f, f)
        acc += amp * a
        amp *= 0.5
    acc = (acc - acc.min()) / (acc.max() - acc.min())
    c = np.clip((acc - 0.45) * 2.2, 0, 1)[..., None]
    sky = np.array([0.35, 0.55, 0.9], np.float32)
    img = sky * (1 - c) + np.array([1, 1, 1], np.float32) * c
    return (np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)


def block_textures(seed=11):
    """Six 160x160 block-face textures (16x16 pixel art, 10x nearest upscale) in the PNG kinds the reference's
    scenes/minecraft_world/tex/*.png use: 2 / 4 / 8-bit palettes, RGBA, grey and RGB.  Written with PIL."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    out = os.path.join(ROOT, "scenes", "blocks")
    os.makedirs(out, exist_ok=True)
    base = [(0.35, 0.6, 0.25), (0.5, 0.38, 0.25), (0.55, 0.55, 0.58), (0.8, 0.75, 0.5), (0.25, 0.4, 0.75), (0.7, 0.3, 0.3)]
    kinds = [("P", 4), ("P", 16), ("P", 256), ("RGBA", 0), ("L", 0), ("RGB", 0)]
    for i, ((mode, colours), tone) in enumerate(zip(kinds, base)):
        cell = rng.random((16, 16, 1)).astype(np.float32)
        img = np.clip(np.array(tone, np.float32) * (0.6 + 0.7 * cell) + 0.08 * rng.random((16, 16, 3)), 0, 1)
        img[0, :] *= 0.7
        img[:, 0] *= 0.7      # a darker seam along two edges of every block face
        rgb = (np.kron(img, np.ones((10, 10, 1), np.float32)) * 255 + 0.5).astype(np.uint8)
        im = Image.fromarray(rgb)
        if mode == "P":
            im = im.quantize(colours)
        elif mode == "RGBA":
            im = im.convert("RGBA")
            a = np.asarray(im).copy()
            a[::20, ::20, 3] = 0      # (alpha is dropped by the loader: lodepng LCT_RGB, texture.cpp:86)
            im = Image.fromarray(a)
        elif mode == "L":
            im = im.convert("L")
        path = os.path.join(out, "block%d.png" % i)
        im.save(path, bits=2 if colours == 4 else (4 if colours == 16 else 8)) if mode == "P" else im.save(path)
        print(path, mode, colours, os.path.getsize(path))


def main():
    block_textures()
    from raytracer_amd import _ffi
    host = _ffi.load_host()
    for name, img in (("tex_bricks.png", bricks()), ("tex_clouds.png", clouds())):
        img = np.ascontiguousarray(img)
        path = os.path.join(ROOT, "scenes", name)
        rc = host.rth_write_png(os.fsencode(path), img.ctypes.data, img.shape[1], img.shape[0], 3)
        print(path, "ok" if rc == 0 else "FAILED", os.path.getsize(path))


if __name__ == "__main__":
    main()
