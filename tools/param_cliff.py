#!/usr/bin/env python3
"""What the NUMBER of scene parameters costs (vector.hpp:185-191: the reference differentiates with respect to any number of
Vector<T,3,true>): fwd+bwd frames of 512 x 512 x 64, depth 8 (config 3's frame) and the reference's default roulette, on scenes
of 4 ... 64 parameters -- kernels' time per frame (HIP events), frames in stream order (wall), which route rendered them, and
the same room with 4 parameters beside every many-parameter scene.  python tools/param_cliff.py [scene ...]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
import torch

names = sys.argv[1:] or ["cornell", "cornell_walls", "cornell_shapes", "params4of16", "params16", "params4of32", "params32",
                         "params4of64", "params64"]
dev = torch.device("cuda", 0)
print(f"{'scene':>16} {'P':>3} {'mode':>10} {'frame_ms':>9} {'kernels_ms':>10} {'launches':>8} {'Gray/s':>7}  kernels")
for name in names:
    scene = pkg.scene_by_name(name)
    r = pkg.HipRenderer(0)
    r.set_specialisation(pkg.SPECIALISE_NOW)
    r.upload_scene(scene)
    cam = pkg.cornell_camera(512, 512)
    out = torch.zeros((512, 512, 3), dtype=torch.float32, device=dev)
    grad = torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev)
    for mode, kw in (("d8", dict(min_bounces=8, absorb=1.0)), ("b1p0.5", dict(min_bounces=1, absorb=0.5))):
        rp = pkg.RenderParams(spp=64, seed=1, flags=pkg.RENDER_SERIAL if hasattr(pkg, "RENDER_SERIAL") else 0, **kw)
        for backward in (True, False):
            for _ in range(30):
                r.render_device(cam, rp, out.data_ptr(), grad.data_ptr() if backward else 0, backward=backward)
            r.synchronize()
            n = 40
            t0 = time.perf_counter()
            for _ in range(n):
                r.render_device(cam, rp, out.data_ptr(), grad.data_ptr() if backward else 0, backward=backward)
            r.synchronize()
            frame = (time.perf_counter() - t0) / n * 1e3
            best = None
            for _ in range(5):
                r.render_device(cam, rp, out.data_ptr(), grad.data_ptr() if backward else 0, backward=backward)
                st = r.render_device(cam, rp, out.data_ptr(), grad.data_ptr() if backward else 0, backward=backward, timing=True)
                ker = sum(v["ms"] for v in st["kernels"].values())
                best = ker if best is None else min(best, ker)
            nl = sum(v["launches"] for v in st["kernels"].values())
            ks = " ".join(f"{k}={v['ms']:.3f}" for k, v in st["kernels"].items() if v["ms"] > 0)
            print(f"{name:>16} {scene.n_params:>3} {mode + ('' if backward else ' fwd'):>10} {frame:9.3f} {best:10.3f} {nl:8d} {st['segments'] / frame / 1e6:7.1f}  {ks}", flush=True)
    r.close()
