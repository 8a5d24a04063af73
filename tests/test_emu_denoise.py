import emu_ffi
import denoise_checks as dc


def test_denoise_matches_cpu_restatement():
    dc.check_denoise(emu_ffi.EmuRenderer(), 96, 64, tol=1e-6)
