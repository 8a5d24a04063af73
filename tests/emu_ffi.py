"""Loader for tests/_emu/librtcore_emu.so — the single-threaded HOST EMULATION build of
raytracer_amd/csrc/rtcore.hip (g++ -DRT_EMU).  CPU-side unit tests use it to exercise the queue
logic and host orchestration of the integrator where no GPU exists.  It is test infrastructure:
never shipped in raytracer_amd/lib, never loaded by the package, not a fallback."""
import ctypes as C
import os

from raytracer_amd import Renderer, _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "_emu", "librtcore_emu.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build()")
        _lib = _ffi.bind_core(C.CDLL(path))
    return _lib


def EmuRenderer():
    return Renderer(0, _lib=load())
