#!/usr/bin/env python3
"""One render through both device routes in f64 against the oracle, row by row where the ray counts differ.
Usage: tools/diag_case.py scene W H spp min_bounces absorb seed [unbiased]"""
import dataclasses, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()
name, w, h, spp, b, p, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
unbiased = len(sys.argv) > 8
scene = pkg.scene_by_name(name)
cam = pkg.cornell_camera(w, h) if not name.startswith("random") else pkg.Camera(w, h).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=seed)
r = pkg.HipRenderer(0); r.upload_scene(scene)
o = oracle.render(scene, cam, rp, backward=True, unbiased=unbiased, zero_dir_miss=unbiased)
sc = np.abs(o["grads"]).max()
print("oracle", o["stats"])
for tag, nb in (("default route", 0), ("wavefront", 1)):
    _, g, st = r.render(cam, dataclasses.replace(rp, bounces_per_launch=nb), backward=True, f64=True, unbiased=unbiased)
    print(tag, "segments", st["segments"], "capped", st["capped_paths"], "grad dev", float(np.abs(g - o["grads"]).max() / sc), "path launches", st["kernels"]["path"]["launches"])
for y in range(h):
    rps = dataclasses.replace(rp, shard=y, n_shards=h, band_rows=1)
    oo = oracle.render(scene, cam, rps, backward=True, unbiased=unbiased, zero_dir_miss=unbiased)
    res = []
    for nb in (0, 1):
        _, g, st = r.render(cam, dataclasses.replace(rps, bounces_per_launch=nb), backward=True, f64=True, unbiased=unbiased)
        res.append((st["segments"], float(np.abs(g - oo["grads"]).max() / max(1e-300, np.abs(oo["grads"]).max()))))
    if res[0][0] != oo["stats"]["segments"] or res[1][0] != oo["stats"]["segments"]:
        print("row", y, "oracle", oo["stats"]["segments"], "deepest", oo["stats"]["deepest"], "default", res[0], "wavefront", res[1])
