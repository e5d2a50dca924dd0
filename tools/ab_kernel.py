#!/usr/bin/env python3
"""Interleaved A/B of builds of libdrt_hip on ONE kernel of a render: ms per step of st["kernels"][AB_KERNEL] (HIP events), median of
AB_ROUNDS rounds, the builds alternating in one process.
  AB_SCENE=cornell_specular AB_SIZE=2048 AB_SPP=16 AB_DEPTH=16 AB_KERNEL=path python tools/ab_kernel.py lib1.so lib2.so ...
  (config 5's shape; AB_KERNEL=backward AB_SCENE=mesh160x160fall: K6 with an albedo per face; AB_UNBIASED=1, AB_BACKWARD=0, AB_F64=1)"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import __graft_entry__ as e

pkg = e.load_package()
libs = sys.argv[1:]
env = os.environ.get
size, spp, depth = int(env("AB_SIZE", "512")), int(env("AB_SPP", "64")), int(env("AB_DEPTH", "8"))
kernel, rounds = env("AB_KERNEL", "path"), int(env("AB_ROUNDS", "7"))
kw = dict(backward=env("AB_BACKWARD", "1") != "0", unbiased=env("AB_UNBIASED", "0") == "1", f64=env("AB_F64", "0") == "1")
for scene_name in env("AB_SCENE", "cornell").split(","):
    scene = pkg.scene_by_name(scene_name)
    cam = pkg.cornell_camera(size, size)
    rp = pkg.RenderParams(spp=spp, seed=1, min_bounces=depth, absorb=1.0)
    rs = [pkg.HipRenderer(0, lib_path=os.path.abspath(l)) for l in libs]
    for r in rs:
        r.set_specialisation(pkg.SPECIALISE_NOW)
        r.upload_scene(scene)
        for _ in range(2):
            r.render(cam, rp, **kw)
    res = {l: [] for l in libs}
    for rnd in range(rounds):
        for l, r in zip(libs, rs):
            _, _, st = r.render(cam, rp, timing=True, **kw)
            res[l].append(st["kernels"][kernel]["ms"])
    print(scene_name, f"{size}x{size}x{spp} depth {depth}", kernel, "  ".join(f"{os.path.basename(l)} {np.median(res[l]):.4f}" for l in libs), flush=True)
    for r in rs:
        r.close()
