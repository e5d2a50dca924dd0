#!/usr/bin/env python3
"""Host-buffer renders of config 3's frame: drt_hip_render (synchronous) against drt_hip_render_async / drt_hip_wait
with two to four frames in flight; where the time of the pipelined loop goes (submit, wait)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
r.upload_scene(pkg.cornell_box())
cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
for _ in range(3):
    r.render(cam, rp, backward=True)
n = 20
t0 = time.perf_counter()
for _ in range(n):
    r.render(cam, rp, backward=True)
print(f"drt_hip_render: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per frame")
r.wait(r.render_async(cam, rp, backward=True))
for depth in (2, 3, pkg.FRAMES_IN_FLIGHT):
    ts, tw = [], []
    bufs = [(np.zeros((512, 512, 3), np.float32), np.zeros((4, 3), np.float64)) for _ in range(pkg.FRAMES_IN_FLIGHT)]
    n = 40
    flying = []
    t0 = time.perf_counter()
    for it in range(n):
        a = time.perf_counter()
        flying.append(r.render_async(cam, rp, backward=True, img_out=bufs[it % len(bufs)][0], grads_out=bufs[it % len(bufs)][1]))
        b = time.perf_counter()
        if len(flying) == depth:
            r.wait(flying.pop(0), want_stats=False)
        c = time.perf_counter()
        ts.append(b - a); tw.append(c - b)
    while flying:
        r.wait(flying.pop(0), want_stats=False)
    print(f"render_async + wait, {depth} frames in flight: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per frame (submit {np.median(ts) * 1e3:.3f}, wait {np.median(tw) * 1e3:.3f})")
