#!/usr/bin/env python3
"""Replay of tools/fuzz_reference.py 36000 621, case 19619 (random7, unbiased, absorb 0.2, min_bounces 6: 8 paths reach the library's 64
vertices): the device's three ways of rendering it -- the one-launch kernel's general form, its column form (DRT_HIP_GEN_ABOVE=8 in
another process), the queue wavefront -- against each other and against the reference."""
import dataclasses, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()
rs = np.random.RandomState(621)
scenes = ["cornell", "cornell_specular", "cornell_walls", "cornell_emissive_wall", "cornell_mirror_wall", "random3", "random7",
          "random19", "mesh6x8", "mesh10x12f5", "cornell_shapes", "params12", "params20", "mesh10x12f12", "cornell_disc_box", "cornell_disc",
          "cornell_coslobe", "cornell_coslobe_disc"]
for case in range(19620):
    name = scenes[rs.randint(len(scenes))]
    mesh = name.startswith("mesh")
    w, h = (int(rs.randint(6, 40)), int(rs.randint(6, 32))) if mesh else (int(rs.randint(8, 96)), int(rs.randint(8, 72)))
    fixed = rs.rand() < 0.5
    b = int(rs.randint(1, 7))
    p = 1.0 if fixed else float(rs.choice([0.2, 0.35, 0.5, 0.8]))
    spp = int(rs.randint(1, 5 if mesh else 13)); seed = int(rs.randint(1 << 30))
    unbiased = rs.rand() < 0.35
    adjoint = rs.uniform(0.2, 1.5, (h, w, 3)).astype(np.float32) if rs.rand() < 0.4 else None
print("case", case, name, w, h, "spp", spp, "b", b, "p", p, "seed", seed, "unbiased", unbiased, "adjoint", adjoint is not None)
scene = pkg.scene_by_name(name)
cam = pkg.cornell_camera(w, h) if not name.startswith("random") else pkg.Camera(w, h).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=seed)
ref = oracle.render_reference(scene, cam, rp, backward=True, adjoint=adjoint, tracer_mode=2 if unbiased else 0, zero_dir_miss=unbiased)
r = pkg.HipRenderer(0); r.upload_scene(scene)
_, g, st = r.render(cam, rp, backward=True, f64=True, unbiased=unbiased, adjoint=adjoint)
_, q, sq = r.render(cam, dataclasses.replace(rp, bounces_per_launch=1), backward=True, f64=True, unbiased=unbiased, adjoint=adjoint)
sc = np.abs(ref["grads"]).max()
print("route", "path" if st["kernels"]["path"]["launches"] else "queue", "GEN_ABOVE", os.environ.get("DRT_HIP_GEN_ABOVE"), "segments", st["segments"], sq["segments"], ref["stats"]["segments"],
      "capped", st["capped_paths"], sq["capped_paths"])
print("one-launch vs reference %.3e   queue vs reference %.3e   one-launch vs queue %.3e" % (np.abs(g - ref["grads"]).max() / sc, np.abs(q - ref["grads"]).max() / sc, np.abs(g - q).max() / sc))
