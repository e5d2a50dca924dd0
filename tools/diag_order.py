#!/usr/bin/env python3
"""The Cornell box with its parameters and materials declared in another order: the gradients (per NAME) must not move."""
import itertools, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()

def permuted(order):
    base = pkg.cornell_box()
    s = pkg.Scene()
    names = ["red", "green", "white", "emission"]
    vals = dict(zip(names, base.params))
    idx = {}
    for n in order:
        idx[n] = s.parameter(vals[n], True, n)
    mats = {}
    for n in order:
        if n != "emission":
            mats[n] = s.diffuse(idx[n])
    em = s.area_emitter(idx["emission"])
    for (t, m, ee, p) in base.shapes:
        mname = None if m < 0 else base.param_names[base.materials[m][1]]
        mat = -1 if m < 0 else mats[mname]
        s.shapes.append((t, mat, em if ee >= 0 else -1, p))
    return s, idx

cam = pkg.cornell_camera(64, 48)
rp = pkg.RenderParams(spp=8, min_bounces=4, absorb=1.0, seed=1)
r = pkg.HipRenderer(0)
ref = None
for order in [["red", "green", "white", "emission"], ["white", "red", "green", "emission"], ["emission", "white", "green", "red"], ["green", "emission", "red", "white"]]:
    s, idx = permuted(order)
    r.upload_scene(s)
    img, g, st = r.render(cam, rp, backward=True)
    img64, g64, st64 = r.render(cam, rp, backward=True, f64=True)
    o = oracle.render(s, cam, rp, backward=True)
    by = {n: g[idx[n]] for n in order}; by64 = {n: g64[idx[n]] for n in order}; byo = {n: o["grads"][idx[n]] for n in order}
    print(order)
    for n in ["red", "green", "white", "emission"]:
        print("  ", n.ljust(9), "f32", np.array2string(by[n], precision=4), "f64", np.array2string(by64[n], precision=4), "oracle", np.array2string(byo[n], precision=4),
              "ERR" if np.abs(by[n] - byo[n]).max() > 1e-3 * np.abs(byo[n]).max() else "")
