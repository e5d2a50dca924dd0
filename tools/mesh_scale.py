import sys, os
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
r.upload_scene(pkg.scene_by_name("mesh160x160"))
cam = pkg.cornell_camera(512, 512)
for spp in (1, 4, 16, 64):
    rp = pkg.RenderParams(spp=spp, min_bounces=8, absorb=1.0, seed=1)
    for _ in range(2):
        r.render(cam, rp, backward=True)
    _, _, st = r.render(cam, rp, backward=True, timing=True)
    k = st["kernels"]
    print(f"spp {spp:3d}: rays {st['segments']:10d}  mesh {k['intersect_mesh']['ms']:7.3f} ms  intersect {k['intersect']['ms']:6.3f}  shade {k['shade']['ms']:6.3f}  -> mesh ns/ray {k['intersect_mesh']['ms']*1e6/st['segments']:.3f}")
