#!/usr/bin/env python3
"""Diagnosis of a long-fuzz finding: cornell_specular 30x26 spp 6 b0 p0.1 seed 190848942, unbiased, f64: the one-launch kernel
and the queue wavefront count different numbers of segments (paths reach the depth cap of 64)."""
import dataclasses, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()
scene = pkg.scene_by_name("cornell_specular"); cam = pkg.cornell_camera(30, 26)
r = pkg.HipRenderer(0); r.upload_scene(scene)
for md in (0, 64, 63, 62, 60, 56, 48, 40, 32, 24, 20, 7):
    rp = pkg.RenderParams(spp=6, min_bounces=0, absorb=0.1, seed=190848942, max_depth=md)
    o = oracle.render(scene, cam, rp, backward=True, unbiased=True, zero_dir_miss=True)
    a = r.render(cam, rp, backward=True, f64=True, unbiased=True)
    q = r.render(cam, dataclasses.replace(rp, bounces_per_launch=1), backward=True, f64=True, unbiased=True)
    sc = np.abs(o["grads"]).max()
    print(f"max_depth {md}: oracle {o['stats']['segments']} (deepest path {o['stats']['max_vertices']})  one-launch {a[2]['segments']} capped {a[2]['capped_paths']} "
          f"grad dev {np.abs(a[1] - o['grads']).max() / sc:.2e}   queue {q[2]['segments']} capped {q[2]['capped_paths']} grad dev {np.abs(q[1] - o['grads']).max() / sc:.2e}")
