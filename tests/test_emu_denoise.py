import emu_ffi
import denoise_checks as dc


def test_denoise_matches_cpu_restatement():
    dc.check_denoise(emu_ffi.EmuRenderer(), 96, 64, tol=1e-6)


def test_denoise_noalloc_logic_on_emulation():
    """rt_denoise_noalloc (for callers that cannot allocate): same result as rt_denoise, twice in a row with a smaller
    second image, and a clear error beyond the fixed working array.  (The GPU test is tests/test_zz_no_alloc.py.)"""
    import numpy as np
    import pytest
    import emu_ffi
    from denoise_checks import synthetic_frame
    r = emu_ffi.EmuRenderer()
    for w, h in ((96, 64), (40, 30)):
        _, noisy, albedo, normal = synthetic_frame(w, h, seed=2)
        want = r.Denoise(noisy, albedo, normal)
        out = np.full_like(noisy, -7.0)
        before = r.alloc_count()
        r.denoise_noalloc(noisy.ctypes.data, out.ctypes.data, w, h, albedo.ctypes.data, normal.ctypes.data)
        assert r.alloc_count() == before and np.array_equal(out, want)
    with pytest.raises(RuntimeError, match="exceeds the fixed working array"):
        r.denoise_noalloc(noisy.ctypes.data, out.ctypes.data, 2048, 2048)
