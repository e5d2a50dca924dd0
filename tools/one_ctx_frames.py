#!/usr/bin/env python3
"""One context, device-pointer renders back to back (no communicator): ms per frame."""
import sys, time
sys.path.insert(0, '.')
import torch
import __graft_entry__ as e
pkg = e.load_package()
scene = pkg.scene_by_name("cornell"); cam = pkg.cornell_camera(512, 512)
dev = torch.device("cuda", 0)
r = pkg.HipRenderer(0)
r.upload_scene(scene)
out = torch.zeros((512, 512, 3), dtype=torch.float32, device=dev)
grads = torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev)
for b, p in ((8, 1.0), (1, 0.5)):
    rp = pkg.RenderParams(spp=64, min_bounces=b, absorb=p, seed=1)
    for rep in range(3):
        for _ in range(4):
            r.render_device(cam, rp, out.data_ptr(), grads.data_ptr())
        r.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            r.render_device(cam, rp, out.data_ptr(), grads.data_ptr())
        r.synchronize()
        print(f"-b {b} -p {p}: {(time.perf_counter() - t0) / 40 * 1e3:.4f} ms per frame")
