import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); r = pkg.HipRenderer(0)
sc = pkg.scene_by_name("random3"); r.upload_scene(sc)
cam = pkg.cornell_camera(512, 512); rp = pkg.RenderParams(spp=32, min_bounces=2, absorb=0.25, seed=41)
_, g, _ = r.render(cam, rp, backward=True); _, g64, _ = r.render(cam, rp, backward=True, f64=True)
err = np.abs(g - g64); p = int(np.unravel_index(err.argmax(), err.shape)[0])
print("grads f64\n", g64, "\nabs err\n", err, "\nworst param", p, sc.param_names[p], sc.materials)
i32, gi32, _ = r.render_gradient_image(cam, rp, p); i64, gi64, _ = r.render_gradient_image(cam, rp, p, f64=True)
d = np.abs(gi32.astype(np.float64) - gi64).max(-1)
idx = np.argsort(d.ravel())[::-1][:8]
print("gradient image: total abs diff", d.sum() * rp.spp, "top pixels:")
for k in idx:
    y, x = divmod(int(k), 512)
    print((x, y), "diff", d[y, x] * rp.spp, "gimg64", gi64[y, x] * rp.spp, "gimg32", gi32[y, x] * rp.spp, "img64", i64[y, x], "img32", i32[y, x])
print("gimg64 abs: max", np.abs(gi64).max() * rp.spp, "mean", np.abs(gi64).mean() * rp.spp, "sum", gi64.sum((0, 1)) * rp.spp)
