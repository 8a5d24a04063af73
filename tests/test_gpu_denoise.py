import pytest

import denoise_checks as dc

pytestmark = pytest.mark.gpu


def test_denoise_matches_cpu_restatement(gpu_renderer):
    dc.check_denoise(gpu_renderer, 640, 360, tol=2e-5)
