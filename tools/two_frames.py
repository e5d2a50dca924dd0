#!/usr/bin/env python3
"""Do two frames on two streams overlap (the tail of one k_path grid filled by the start of the next)?  Two contexts on
device 0, device-pointer renders enqueued alternately, against one context rendering the same number of frames."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import __graft_entry__ as e
pkg = e.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
scene = pkg.scene_by_name(name); cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
dev = torch.device("cuda", 0)
rs = [pkg.HipRenderer(0) for _ in range(3)]
outs = [torch.zeros((512, 512, 3), dtype=torch.float32, device=dev) for _ in rs]
grads = [torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in rs]
for r in rs:
    r.upload_scene(scene)
def run(n_ctx, frames=40):
    for i in range(4):
        rs[i % n_ctx].render_device(cam, rp, outs[i % n_ctx].data_ptr(), grads[i % n_ctx].data_ptr())
    for r in rs: r.synchronize()
    t0 = time.perf_counter()
    for i in range(frames):
        k = i % n_ctx
        rs[k].render_device(cam, rp, outs[k].data_ptr(), grads[k].data_ptr())
    for r in rs: r.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3
for rep in range(3):
    print(name, "ms per frame: one stream %.4f   two streams %.4f   three streams %.4f" % (run(1), run(2), run(3)))
print("grads equal:", bool((grads[0] == grads[1]).all().item()))
