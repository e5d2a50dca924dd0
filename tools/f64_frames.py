#!/usr/bin/env python3
"""N frames of config 3 (Cornell 512 x 512 x 64, depth 8, fwd+bwd) in the f64 mode (DRT_RENDER_F64: the reference's own precision,
render.cpp:22), frames in stream order -- the workload behind profiles/r06_f64_*: time it, or run it under rocprofv3.
python tools/f64_frames.py [frames] [scene]"""
import sys, time
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
scene = pkg.scene_by_name(sys.argv[2] if len(sys.argv) > 2 else "cornell")
r = pkg.HipRenderer(0)
r.set_specialisation(pkg.SPECIALISE_NOW)
r.upload_scene(scene)
cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
for _ in range(2):
    r.render(cam, rp, backward=True, f64=True)
best = None
t0 = time.perf_counter()
for _ in range(n):
    _, _, st = r.render(cam, rp, backward=True, f64=True, timing=True)
    ms = st["kernels"]["path"]["ms"]
    best = ms if best is None else min(best, ms)
print(f"f64 frames: {n}, k_path best {best:.4f} ms, segments {st['segments']}, {st['segments'] / best / 1e6:.1f} Gray/s in-kernel, "
      f"wall per frame {(time.perf_counter() - t0) / n * 1e3:.3f} ms (host buffers, timing on)")
