import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as e
from conftest import load_golden, case_inputs
pkg = e.load_package(); r = pkg.HipRenderer(0)
for name in ("g7_random3_40x30x6", "g8_random8_36x36x6_d5"):
    g = load_golden(name)
    scene, cam, rp, adj = case_inputs(pkg, g["case"])
    r.upload_scene(scene)
    img, grads, st = r.render(cam, rp, backward=True, adjoint=adj)
    img64, grads64, st64 = r.render(cam, rp, backward=True, adjoint=adj, f64=True)
    err = np.abs(grads - g["grads"]); scale = np.abs(g["grads"]).max()
    print(name, "f32 rel err", err.max() / scale, "f64 rel err", np.abs(grads64 - g["grads"]).max() / scale, "segments", st["segments"], st64["segments"], int(g["segments"]))
    p = int(np.unravel_index(err.argmax(), err.shape)[0])
    i32, gi32, _ = r.render_gradient_image(cam, rp, p, adjoint=adj); i64, gi64, _ = r.render_gradient_image(cam, rp, p, adjoint=adj, f64=True)
    d = np.abs(gi32.astype(np.float64) - gi64).max(-1) * rp.spp
    idx = np.argsort(d.ravel())[::-1][:5]
    print(" worst param", p, "gradient image diffs (abs, per pixel sum): total", d.sum(), "top:")
    for k in idx:
        y, x = divmod(int(k), cam.width)
        print("  ", (x, y), "diff", d[y, x], "gimg64", gi64[y, x] * rp.spp, "gimg32", gi32[y, x] * rp.spp, "img64", i64[y, x], "img32", i32[y, x])
    for bpl in (1,):
        import dataclasses
        _, g1, s1 = r.render(cam, dataclasses.replace(rp, bounces_per_launch=bpl), backward=True, adjoint=adj)
        print(" queue route rel err", np.abs(g1 - g["grads"]).max() / scale, s1["segments"])
