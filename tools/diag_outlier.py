#!/usr/bin/env python3
"""Localise an f32-vs-f64 gradient difference to pixels (works for the unbiased backward too, which
has no gradient image): per-row, then per-pixel adjoint masks.
Usage: tools/diag_outlier.py <scene> <w> <h> <spp> <min_bounces> <absorb> <seed> [unbiased]"""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
name, w, h, spp, b, p, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
unb = len(sys.argv) > 8 and sys.argv[8] == "unbiased"
sc = pkg.scene_by_name(name)
r.upload_scene(sc)
cam = pkg.cornell_camera(w, h)
rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=seed)


def both(adj):
    _, g, _ = r.render(cam, rp, backward=True, adjoint=adj, unbiased=unb)
    _, g64, _ = r.render(cam, rp, backward=True, adjoint=adj, f64=True, unbiased=unb)
    return g, g64


g, g64 = both(None)
scale = np.abs(g64).max()
print("full frame: rel err", np.abs(g - g64).max() / scale, "\nf64\n", g64, "\nf32-f64\n", g - g64)
rows = []
for y in range(h):
    adj = np.zeros((h, w, 3), np.float32)
    adj[y] = 1
    a, b64 = both(adj)
    rows.append(np.abs(a - b64).max())
rows = np.array(rows)
order = np.argsort(rows)[::-1][:3]
print("worst rows", [(int(y), rows[y] / scale) for y in order])
y = int(order[0])
px = []
for x in range(w):
    adj = np.zeros((h, w, 3), np.float32)
    adj[y, x] = 1
    a, b64 = both(adj)
    px.append((np.abs(a - b64).max(), np.abs(b64).max()))
px = np.array(px)
for x in np.argsort(px[:, 0])[::-1][:4]:
    print("pixel", (int(x), y), "abs diff / total scale", px[x, 0] / scale, " pixel's own |grad| / total", px[x, 1] / scale)
    adj = np.zeros((h, w, 3), np.float32)
    adj[y, x] = 1
    a, b64 = both(adj)
    print("   f32", a.ravel()[:6], "\n   f64", b64.ravel()[:6])
