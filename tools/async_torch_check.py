import sys, time
import numpy as np
sys.path.insert(0, '.')
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode != "plain":
    import torch
    if mode == "torchcuda":
        x = torch.zeros((512, 512, 3), device="cuda")
        torch.cuda.synchronize()
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
sync = (time.perf_counter() - t0) / n * 1e3
r.wait(r.render_async(cam, rp, backward=True))
t0 = time.perf_counter()
prev = None
for _ in range(n):
    h = r.render_async(cam, rp, backward=True)
    if prev is not None:
        r.wait(prev, want_stats=False)
    prev = h
r.wait(prev)
print(mode, f"sync {sync:.3f} ms, pipelined {(time.perf_counter() - t0) / n * 1e3:.3f} ms")
