"""The loop the path exists for (/root/reference/README.md:88-101: render -> loss -> backward -> parameter step), through
the C ABI on the device: recover the red wall's albedo (0.5, 0, 0) of the reference's scene from (0.2, 0.2, 0.2) by gradient
descent on an L2 image loss -- tools/fit_albedo.py -- with drt_hip_render and with drt_hip_render_async / drt_hip_wait.
The zero channels of the target exercise the gradient with respect to a colour channel that IS zero (k_path's counters,
csrc/drt_path.h: Tangents).  Needs a real MI355X."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


@pytest.mark.parametrize("use_async", [False, True])
def test_gradient_descent_recovers_the_albedo(pkg, use_async):
    import fit_albedo
    size, spp, steps = 128, 16, 60
    render = fit_albedo.DeviceRender(pkg, size, spp, 8, use_async)
    try:
        rgb, hist = fit_albedo.fit(render, len(render.params0), 0, np.array([0.2, 0.2, 0.2]), steps, spp, size * size * 3)
    finally:
        render.r.close()
    assert np.abs(rgb - np.array([0.5, 0.0, 0.0])).max() <= 1e-2, rgb
    assert render.calls == 2 * steps
