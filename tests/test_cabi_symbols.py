"""The C-ABI libraries load and export every symbol their headers declare (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

from conftest import ROOT
from raytracer_amd import _ffi


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_rtcore_exports_every_declared_symbol():
    names = declared("rtcore.h", "rt_")
    assert len(names) >= 23
    lib = C.CDLL(os.path.join(_ffi.LIB_DIR, "librtcore.so"))
    for n in names:
        assert hasattr(lib, n), f"librtcore.so does not export {n}"
    assert sorted(_ffi.CORE_SYMBOLS) == names


def test_rthost_exports_every_declared_symbol():
    names = declared("rthost.h", "rth_")
    lib = C.CDLL(os.path.join(_ffi.LIB_DIR, "librthost.so"))
    for n in names:
        assert hasattr(lib, n), f"librthost.so does not export {n}"
    assert sorted(_ffi.HOST_SYMBOLS) == names


def test_struct_sizes_match_header(tmp_path):
    """ctypes mirrors vs sizeof() from the C compiler on include/rtcore.h itself."""
    import subprocess
    names = ["rt_ray", "rt_hit", "rt_photon", "rt_node", "rt_bvh_node", "rt_mesh", "rt_material", "rt_texmap", "rt_light",
             "rt_camera", "rt_scene_desc", "rt_render_params", "rt_framebuffer", "rt_counters", "rt_profile"]
    src = '#include <stdio.h>\n#include "rtcore.h"\nint main(void){' + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    c = tmp_path / "sz.c"
    c.write_text(src)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    for n, sz in zip(names, sizes):
        assert C.sizeof(getattr(_ffi, n)) == sz, n


def test_product_path_fails_loudly_without_gpu_or_library(monkeypatch, tmp_path):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from raytracer_amd import Renderer
    with pytest.raises(RuntimeError):
        Renderer(0)  # rt_context_create reports no usable device; nothing falls back to a CPU path
    monkeypatch.setattr(_ffi, "LIB_DIR", str(tmp_path))
    monkeypatch.setattr(_ffi, "_core", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ffi.load_core()


def test_default_params_match_c_side():
    lib = _ffi.bind_core(C.CDLL(os.path.join(_ffi.LIB_DIR, "librtcore.so")))
    p = _ffi.rt_render_params()
    lib.rt_render_params_default(C.byref(p))
    q = _ffi.default_params()
    for name, _ in _ffi.rt_render_params._fields_:
        a, b = getattr(p, name), getattr(q, name)
        if name == "reserved":
            continue
        assert a == b or abs(a - b) < 1e-7, name
