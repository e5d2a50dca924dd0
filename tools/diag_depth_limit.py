#!/usr/bin/env python3
"""Renders whose deepest trace() stands exactly at the library's depth limit of 64 vertices, unbiased operator, f64: where the
reference's roulette ends such a path AT depth 64, the device and the reference must stay in step (the draw counts as drawn).
Usage: tools/diag_depth_limit.py [lib.so]"""
import dataclasses, os, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()
r = pkg.HipRenderer(0, lib_path=os.path.abspath(sys.argv[1])) if len(sys.argv) > 1 else pkg.HipRenderer(0)
for name, seed in (("cornell", 253), ("cornell", 1246), ("cornell", 1438), ("cornell_specular", 252), ("cornell_specular", 289), ("cornell_specular", 423)):
    scene = pkg.scene_by_name(name); cam = pkg.cornell_camera(12, 10)
    rp = pkg.RenderParams(spp=4, min_bounces=0, absorb=0.17, seed=seed)
    o = oracle.render(scene, cam, rp, backward=True, unbiased=True, zero_dir_miss=True)
    r.upload_scene(scene)
    a = r.render(cam, rp, backward=True, f64=True, unbiased=True)
    q = r.render(cam, dataclasses.replace(rp, bounces_per_launch=1), backward=True, f64=True, unbiased=True)
    sc = np.abs(o["grads"]).max()
    print(f"{name} seed {seed}: oracle {o['stats']['segments']} deepest {o['stats']['deepest']}  one-launch {a[2]['segments']} capped {a[2]['capped_paths']} "
          f"grad dev {np.abs(a[1] - o['grads']).max() / sc:.2e}   queue {q[2]['segments']} capped {q[2]['capped_paths']} grad dev {np.abs(q[1] - o['grads']).max() / sc:.2e}")
