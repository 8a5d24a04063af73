"""rt_denoise_noalloc (include/rtcore.h): the entry point for callers that cannot allocate.  Called through the C-ABI with
device pointers and checked against rt_denoise — the existing entry point it is a variant of, same kernels — bit for bit,
on two calls in a row, the second with a smaller image (the working array keeps the first call's contents)."""
import numpy as np
import pytest

from denoise_checks import synthetic_frame

pytestmark = pytest.mark.gpu


def _dev(t):
    import torch
    return torch.from_numpy(np.ascontiguousarray(t)).to("cuda:0")


def test_denoise_noalloc_matches_rt_denoise(gpu_renderer):
    import torch
    r = gpu_renderer
    for (w, h, guides) in ((640, 360, True), (200, 120, True), (333, 77, False), (1024, 1024, False)):
        clean, noisy, albedo, normal = synthetic_frame(w, h, seed=w)
        a, n = (albedo, normal) if guides else (None, None)
        want = r.Denoise(noisy, a, n)                       # rt_denoise: host pointers, buffers from the context
        d_in, d_out = _dev(noisy), torch.full((h, w, 3), -7.0, dtype=torch.float32, device="cuda:0")
        d_a = _dev(a) if guides else None
        d_n = _dev(n) if guides else None
        torch.cuda.synchronize()       # (the inputs' producer stream must have finished before the call, include/rtcore.h)
        before = r.alloc_count()
        free_before = torch.cuda.mem_get_info(0)[0]     # hipMemGetInfo: the DEVICE's free memory, whoever allocates
        r.denoise_noalloc(d_in.data_ptr(), d_out.data_ptr(), w, h, d_a.data_ptr() if guides else 0, d_n.data_ptr() if guides else 0)
        # the library's own counter only sees the allocation sites that bump it; the device's free memory sees any hipMalloc
        # (or lazily allocating call) made on the way — it must stand still across the call
        assert torch.cuda.mem_get_info(0)[0] == free_before, "device memory was allocated or freed during rt_denoise_noalloc"
        assert r.alloc_count() == before
        got = d_out.cpu().numpy()
        assert np.array_equal(got, want), f"{w}x{h}: differs from rt_denoise by {np.abs(got - want).max()}"
    # beyond the fixed working array: a clear error, nothing launched
    big_in = torch.zeros((1025, 1024, 3), dtype=torch.float32, device="cuda:0")
    big_out = torch.zeros_like(big_in)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="exceeds the fixed working array"):
        r.denoise_noalloc(big_in.data_ptr(), big_out.data_ptr(), 1024, 1025)
