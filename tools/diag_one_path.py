#!/usr/bin/env python3
"""Diagnosis of a long-fuzz finding: mesh10x12f5 29x28 spp 3 b4 p0.2 seed 66973654, unbiased: device f64 92687 segments, reference 92667."""
import dataclasses, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()
scene = pkg.scene_by_name("mesh10x12f5"); cam = pkg.cornell_camera(29, 28)
rp = pkg.RenderParams(spp=3, min_bounces=4, absorb=0.2, seed=66973654)
r = pkg.HipRenderer(0); r.upload_scene(scene)
orc = oracle.render(scene, cam, rp, backward=True, unbiased=True, zero_dir_miss=True)
for tag, kw in (("f64 unbiased", dict(f64=True, unbiased=True)), ("f32 unbiased", dict(unbiased=True)), ("f64 biased", dict(f64=True))):
    img, g, st = r.render(cam, rp, backward=True, **kw)
    o = orc if "unbiased" in tag else oracle.render(scene, cam, rp, backward=True)
    print(tag, "segments", st["segments"], "oracle", o["stats"]["segments"], "capped", st["capped_paths"],
          "grad dev", float(np.abs(g - o["grads"]).max() / np.abs(o["grads"]).max()), "img dev", float(np.abs(img - o["image"]).max()))
for bpl in (1, 2, 0):
    _, g, st = r.render(cam, dataclasses.replace(rp, bounces_per_launch=bpl), backward=True, f64=True, unbiased=True)
    print("bounces_per_launch", bpl, "segments", st["segments"])
for bp in (256, 1000, 2436):
    _, g, st = r.render(cam, dataclasses.replace(rp, batch_paths=bp), backward=True, f64=True, unbiased=True)
    print("batch_paths", bp, "segments", st["segments"])
bad = []
for y in range(28):
    rps = dataclasses.replace(rp, shard=y, n_shards=28, band_rows=1)
    _, g, st = r.render(cam, rps, backward=True, f64=True, unbiased=True)
    o = oracle.render(scene, cam, rps, backward=True, unbiased=True, zero_dir_miss=True)
    if st["segments"] != o["stats"]["segments"]:
        bad.append(y)
        print("row", y, "device", st["segments"], "oracle", o["stats"]["segments"], "grad dev", float(np.abs(g - o["grads"]).max() / max(1e-300, np.abs(o["grads"]).max())))
print("rows that differ:", bad)
for spp in (1, 2):
    rps = dataclasses.replace(rp, spp=spp)
    _, g, st = r.render(cam, rps, backward=True, f64=True, unbiased=True)
    o = oracle.render(scene, cam, rps, backward=True, unbiased=True, zero_dir_miss=True)
    print("spp", spp, "device", st["segments"], "oracle", o["stats"]["segments"])
# which pixel of row 16: one-hot adjoint images (the gradient is linear in the adjoint)
rps = dataclasses.replace(rp, shard=16, n_shards=28, band_rows=1)
for x in range(29):
    adj = np.zeros((28, 29, 3), dtype=np.float32); adj[16, x] = 1.0
    _, g, st = r.render(cam, rps, backward=True, f64=True, unbiased=True, adjoint=adj)
    o = oracle.render(scene, cam, rps, backward=True, unbiased=True, zero_dir_miss=True, adjoint=adj)
    dev = float(np.abs(g - o["grads"]).max() / max(1e-300, np.abs(o["grads"]).max()))
    if dev > 1e-9:
        print("pixel x =", x, "dev", dev)
        print("device grads\n", g, "\noracle grads\n", o["grads"])
