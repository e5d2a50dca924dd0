#!/usr/bin/env python3
"""Where the time of a SMALL frame goes (the frames of an optimisation loop: tools/fit_albedo.py): wall time per
drt_hip_render call against the kernels' own time, forward and forward + backward, with and without an adjoint image."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
r.upload_scene(pkg.cornell_box())
r.set_specialisation(pkg.SPECIALISE_NOW) if hasattr(r, "set_specialisation") else None
print(f"{'frame':>16} {'mode':>22} {'call_us':>9} {'kernels_us':>11} {'launches':>9}")
for size, spp, depth in ((64, 4, 4), (128, 16, 4), (256, 8, 4), (256, 16, 8)):
    cam = pkg.cornell_camera(size, size)
    rp = pkg.RenderParams(spp=spp, min_bounces=depth, absorb=1.0, seed=3)
    adj = np.full((size, size, 3), 1e-3, dtype=np.float32)
    for mode, kw in (("forward", dict(backward=False)), ("fwd+bwd", dict(backward=True)), ("fwd+bwd, adjoint", dict(backward=True, adjoint=adj)),
                     ("fwd+bwd, L2 target", dict(backward=True, adjoint=adj, loss_l2=True))):
        for _ in range(20):
            r.render(cam, rp, **kw)
        n = 300
        t0 = time.perf_counter()
        for _ in range(n):
            r.render(cam, rp, **kw)
        call = (time.perf_counter() - t0) / n * 1e6
        _, _, st = r.render(cam, rp, timing=True, **kw)
        ker = sum(v["ms"] for v in st["kernels"].values()) * 1e3
        nl = sum(v["launches"] for v in st["kernels"].values())
        print(f"{size:>5}x{size:<5}x{spp:<3} {mode:>22} {call:9.1f} {ker:11.1f} {nl:9d}")
