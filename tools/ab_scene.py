#!/usr/bin/env python3
"""Interleaved A/B of builds of libdrt_hip on config 3's frame of a named scene (AB_SCENE), fixed depth 8 and the default roulette:
k_path ms (HIP events), median of 11 rounds (AB_BACKWARD=0: forward only; AB_UNBIASED=1: the unbiased operator).  python tools/ab_scene.py lib1.so lib2.so ..."""
import sys, os
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
libs = sys.argv[1:]
backward = os.environ.get("AB_BACKWARD", "1") != "0"
unbiased = os.environ.get("AB_UNBIASED", "0") == "1"
for scene_name in os.environ.get("AB_SCENE", "cornell_shapes").split(","):
    scene = pkg.scene_by_name(scene_name); cam = pkg.cornell_camera(512, 512)
    for mode, kw in (("d8", dict(min_bounces=8, absorb=1.0)), ("b1p0.5", dict(min_bounces=1, absorb=0.5))):
        rp = pkg.RenderParams(spp=64, seed=1, **kw)
        rs = [pkg.HipRenderer(0, lib_path=os.path.abspath(l)) for l in libs]
        for r in rs:
            r.set_specialisation(pkg.SPECIALISE_NOW); r.upload_scene(scene)
            for _ in range(3):
                r.render(cam, rp, backward=backward, unbiased=unbiased)
        res = {l: [] for l in libs}
        for rnd in range(11):
            for l, r in zip(libs, rs):
                _, _, st = r.render(cam, rp, backward=backward, unbiased=unbiased, timing=True)
                res[l].append(st["kernels"]["path"]["ms"])
        print(scene_name, mode, "  ".join(f"{os.path.basename(l)} {np.median(res[l]):.4f}" for l in libs), flush=True)
        for r in rs:
            r.close()
