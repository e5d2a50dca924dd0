#!/usr/bin/env python3
"""Interleaved A/B of several builds of libdrt_hip on the glossy scene (config 5's shape at 1024 x 1024 x 32 spp, depth 16,
and a roulette-terminated render): k_path time per launch.  Usage: tools/ab_specular.py lib1.so lib2.so ..."""
import sys, os
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
libs = sys.argv[1:]
scene = pkg.scene_by_name("cornell_specular"); cam = pkg.cornell_camera(1024, 1024)
cases = {"depth 16": pkg.RenderParams(spp=32, min_bounces=16, absorb=1.0, seed=1), "-b 3 -p 0.2": pkg.RenderParams(spp=32, min_bounces=3, absorb=0.2, seed=1)}
rs = []
for l in libs:
    try:
        rs.append(pkg.HipRenderer(0, lib_path=os.path.abspath(l)))
    except Exception as ex:           # (an older ABI)
        print("skip", l, ex); rs.append(None)
for r in rs:
    if r: r.upload_scene(scene)
for name, rp in cases.items():
    res = {l: [] for l in libs}
    for r in rs:
        if r:
            for _ in range(2): r.render(cam, rp, backward=True)
    for rnd in range(5):
        for l, r in zip(libs, rs):
            if not r: continue
            img, g, st = r.render(cam, rp, backward=True, timing=True)
            res[l].append(st["ms_total"] if "ms_total" in st else 0.0)
            k = st["kernels"]
            res[l][-1] = sum(k[n]["ms"] for n in k)
    for l in libs:
        if res[l]:
            print(f"{name:12s} {os.path.basename(l):22s} GPU ms {np.median(res[l]):8.3f}")
