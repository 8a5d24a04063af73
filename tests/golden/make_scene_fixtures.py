#!/usr/bin/env python3
"""Makes tests/golden/scenes/ref_*.xml: the scene DATA of the bundled reference scenes (devinFink/RayTracer scenes/*.xml
and scenes/minecraft_world/mc_world.xml), so that the GPU box — where /root/reference does not exist — can render them
against the oracle ("every bundled scene renders").  Only data travels: XML comments are dropped, Wavefront
meshes the reference checkout does not contain are replaced by this build's seeded procedural meshes, and
image textures by this repository's own generated ones (scenes/tex_*.png, scenes/blocks/*.png, scenes/make_textures.py).
scene3_2.xml is byte-identical to scene3.xml and cornellBox.xml is what tests/golden/scenes/cornell_box.xml restates: both
are covered by those files.  Run in the build container:  python tests/golden/make_scene_fixtures.py"""
import os
import re
import zlib

REF = "/root/reference/RayTracer/scenes"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")
SCENES = ["simple_scene", "project_1_scene", "project_2_scene", "scene3", "scene_4", "scene_5", "weirdhorsething", "scene6",
          "scene_textures", "soft_shadow_scene", "custom", "finalProject", "havingFun", "mc", "volumetric_spiral",
          "minecraft_world/mc_world"]
# stand-ins for the meshes by file stem; anything else (the per-material pieces of the block world) becomes a seeded
# set of textured cubes
MESH = {"teapot": "gen:teapot_like?seed=1&amp;tris=4000", "hollow_teapot": "gen:teapot_like?seed=2&amp;tris=3000",
        "dragon-mesh": "gen:blob?seed=9&amp;tris=5120&amp;size=4",
        "therest": "gen:terrain?seed=21&amp;tris=8192&amp;size=40", "clouds": "gen:blob?seed=22&amp;tris=1280&amp;size=8",
        "water": "gen:terrain?seed=23&amp;tris=512&amp;size=25", "flowers": "gen:blob?seed=24&amp;tris=320&amp;size=3",
        "grassblock": "gen:voxels?seed=25&amp;tris=600", "grass": "gen:terrain?seed=26&amp;tris=2048&amp;size=30",
        "trees": "gen:blob?seed=27&amp;tris=1280&amp;size=10"}


def convert(name):
    text = open(os.path.join(REF, name + ".xml")).read()
    text = re.sub(r"<!--.*?-->", "", text, flags=re.S)

    def mesh(m):
        stem = os.path.splitext(os.path.basename(m.group(1)))[0]
        if stem in MESH:
            return 'name="%s"' % MESH[stem]
        if "minecraft_world" in m.group(1):
            return 'name="gen:voxels?seed=%d&amp;tris=%d"' % (zlib.crc32(stem.encode()) % 100000, 12 * (6 + zlib.crc32(stem.encode()) % 9))
        return 'name="gen:icosphere?seed=5&amp;tris=1280"'
    text = re.sub(r'name="([^"]*\.obj)"', mesh, text)
    text = re.sub(r'texture="(?:scenes/)?bricks\.png"', 'texture="scenes/tex_bricks.png"', text)
    text = re.sub(r'texture="(?:scenes/)?clouds\.png"', 'texture="scenes/tex_clouds.png"', text)
    text = re.sub(r'texture="(?:scenes/)?marble\.png"', 'texture="scenes/tex_bricks.png"', text)   # (absent from the reference: .MISSING_LARGE_BLOBS)
    # the block world's 180 palettised / RGBA / grey 160x160 textures: six generated ones of the same PNG kinds
    text = re.sub(r'texture="scenes/minecraft_world/tex/([^"]*)\.png"',
                  lambda m: 'texture="scenes/blocks/block%d.png"' % (zlib.crc32(m.group(1).encode()) % 6), text)
    text = "\n".join(line.strip() for line in text.splitlines() if line.strip())
    base = os.path.basename(name)
    header = ("<!-- Scene DATA of the reference's bundled scenes/%s.xml (objects, transforms, materials, lights, camera); meshes and\n"
              "     image textures replaced by this repository's procedural ones.  Fixture made by tests/golden/make_scene_fixtures.py;\n"
              "     contains no code. -->\n" % name)
    open(os.path.join(OUT, "ref_" + base + ".xml"), "w").write(header + text + "\n")
    return base


if __name__ == "__main__":
    for s in SCENES:
        print("wrote ref_%s.xml" % convert(s))
