#!/usr/bin/env python3
"""Interleaved A/B of builds of libdrt_hip on the reference's default roulette (-b 1 -p 0.5, 512 x 512 x 64, fwd+bwd): k_path ms."""
import sys, os
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
libs = sys.argv[1:]
scene = pkg.scene_by_name(os.environ.get("AB_SCENE", "cornell")); cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=1, absorb=0.5, seed=1)
rs = [pkg.HipRenderer(0, lib_path=os.path.abspath(l)) for l in libs]
for r in rs:
    r.upload_scene(scene); r.set_specialisation(pkg.SPECIALISE_NOW)
    for _ in range(3):
        r.render(cam, rp, backward=True)
res = {l: [] for l in libs}
for rnd in range(11):
    for l, r in zip(libs, rs):
        _, _, st = r.render(cam, rp, backward=True, timing=True)
        res[l].append(st["kernels"]["path"]["ms"])
for l in libs:
    a = np.array(res[l]); print(os.path.basename(l).ljust(20), f"k_path median {np.median(a):.4f} min {a.min():.4f} ms  segments {st['segments']}")
