#!/usr/bin/env python3
"""Makes tests/golden/scenes/ref_*.xml: the scene DATA of bundled reference scenes (devinFink/RayTracer scenes/*.xml),
so that the GPU box — where /root/reference does not exist — can render them against the oracle ("every
bundled scene renders").  Only data travels: XML comments are dropped, Wavefront meshes the reference checkout does not
contain are replaced by this build's seeded procedural meshes, and image textures by this repository's own
generated ones (scenes/tex_*.png, scenes/make_textures.py).  Run in the build container:  python tests/golden/make_scene_fixtures.py"""
import os
import re

REF = "/root/reference/RayTracer/scenes"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")
SCENES = ["simple_scene", "project_2_scene", "scene_4", "weirdhorsething", "scene6", "scene_textures", "soft_shadow_scene", "custom"]
MESH = {"teapot": "gen:teapot_like?seed=1&amp;tris=4000", "hollow_teapot": "gen:teapot_like?seed=2&amp;tris=3000"}


def convert(name):
    text = open(os.path.join(REF, name + ".xml")).read()
    text = re.sub(r"<!--.*?-->", "", text, flags=re.S)

    def mesh(m):
        stem = os.path.splitext(os.path.basename(m.group(1)))[0]
        return 'name="%s"' % MESH.get(stem, "gen:icosphere?seed=5&amp;tris=1280")
    text = re.sub(r'name="([^"]*\.obj)"', mesh, text)
    text = re.sub(r'texture="(?:scenes/)?bricks\.png"', 'texture="scenes/tex_bricks.png"', text)
    text = re.sub(r'texture="(?:scenes/)?clouds\.png"', 'texture="scenes/tex_clouds.png"', text)
    text = re.sub(r'texture="(?:scenes/)?marble\.png"', 'texture="scenes/tex_bricks.png"', text)   # (absent from the reference: .MISSING_LARGE_BLOBS)
    text = "\n".join(line.rstrip() for line in text.splitlines() if line.strip())
    header = ("<!-- Scene DATA of the reference's bundled scenes/%s.xml (objects, transforms, materials, lights, camera); meshes and\n"
              "     image textures replaced by this repository's procedural ones.  Fixture made by tests/golden/make_scene_fixtures.py;\n"
              "     contains no code. -->\n" % name)
    open(os.path.join(OUT, "ref_" + name + ".xml"), "w").write(header + text + "\n")


if __name__ == "__main__":
    for s in SCENES:
        convert(s)
        print("wrote ref_%s.xml" % s)
