#!/usr/bin/env python3
"""Does the latency-bound BVH walk of one half of the frame overlap with the bandwidth-bound queue kernels of the other?
A group context that lists device 0 several times renders that many row-band shards side by side, one stream each."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
scene = pkg.scene_by_name(sys.argv[1] if len(sys.argv) > 1 else "mesh160x160")
cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1, band_rows=16)
for devs in ([0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
    r = pkg.HipRenderer(devs if len(devs) > 1 else 0)
    r.upload_scene(scene)
    for _ in range(3):
        out = r.render(cam, rp, backward=True)
    t = []
    for _ in range(7):
        t0 = time.perf_counter()
        out = r.render(cam, rp, backward=True)
        t.append(time.perf_counter() - t0)
    print(f"{len(devs)} member(s) on device 0: {1e3 * np.median(t):7.3f} ms per frame (host clock, host buffers), segments {out[2]['segments']}")
    r.close()
