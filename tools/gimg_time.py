#!/usr/bin/env python3
"""Time of a gradient-image render (README.md:142-145) of config 3's frame."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
scene = pkg.scene_by_name("cornell")
r.upload_scene(scene)
cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
for _ in range(3):
    out = r.render_gradient_image(cam, rp, 2)
t = []
for _ in range(7):
    t0 = time.perf_counter(); out = r.render_gradient_image(cam, rp, 2); t.append(time.perf_counter() - t0)
st = out[2]
print(f"gradient image 512x512x64: {1e3 * np.median(t):.3f} ms per call (host buffers), path launches {st['kernels']['path']['launches']}, segments {st['segments']}")
