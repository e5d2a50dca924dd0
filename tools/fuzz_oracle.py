#!/usr/bin/env python3
"""The CPU restatement (oracle/drt_oracle.c) against THE REFERENCE ITSELF (oracle/_ref/ref_harness: the unmodified headers), on
random renders, no GPU involved: image, gradients and ray counts must agree BIT FOR BIT -- both integration operators, adjoint
images, the per-sample squared-error loss, gradient images, triangle meshes, mirror and glossy materials, NaN where the
reference produces NaN.  Runs in the build container (needs /root/reference for the harness).
Usage: tools/fuzz_oracle.py [n_cases] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import __graft_entry__ as e

pkg = e.load_package()
oracle = e.load_oracle()
if not oracle.have_reference():
    print("oracle/_ref/ref_harness is not here: nothing to compare against")
    sys.exit(2)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
scenes = ["cornell", "cornell_specular", "cornell_walls", "cornell_emissive_wall", "cornell_mirror_wall", "random3",
          "random7", "random19", "mesh6x8", "mesh10x12f5"]
t0 = time.time()
n_nan = 0
for case in range(n_cases):
    name = scenes[rs.randint(len(scenes))]
    scene = pkg.scene_by_name(name)
    mesh = name.startswith("mesh")
    w, h = (int(rs.randint(4, 20)), int(rs.randint(4, 16))) if mesh else (int(rs.randint(4, 40)), int(rs.randint(4, 32)))
    cam = pkg.cornell_camera(w, h) if not name.startswith("random") else pkg.Camera(w, h).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
    fixed = rs.rand() < 0.5
    b = int(rs.randint(0, 7))
    p = 1.0 if fixed else float(rs.choice([0.1, 0.2, 0.35, 0.5, 0.8]))
    if fixed and b == 0:
        b = 1
    rp = pkg.RenderParams(spp=int(rs.randint(1, 4 if mesh else 9)), min_bounces=b, absorb=p, seed=int(rs.randint(1 << 30)))
    mode = rs.choice(["biased", "biased", "unbiased", "loss_l2", "gradient image"])
    adjoint = rs.uniform(-0.5, 1.5, (h, w, 3)).astype(np.float32) if (rs.rand() < 0.4 or mode == "loss_l2") else None
    kw_o, kw_r = {}, {}
    if mode == "unbiased":
        kw_o = dict(unbiased=True, zero_dir_miss=True); kw_r = dict(tracer_mode=2, zero_dir_miss=True)
    elif mode == "loss_l2":
        kw_o = dict(loss_l2=True); kw_r = dict(loss_l2=True)
    elif mode == "gradient image":
        gp = int(rs.choice([i for i, g in enumerate(scene.requires_grad) if g]))      # (the reference's .grad() throws for a parameter that is no variable)
        kw_o = dict(grad_image_param=gp); kw_r = dict(grad_image_param=gp)
    ref = oracle.render_reference(scene, cam, rp, backward=True, adjoint=adjoint, **kw_r)
    o = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint, **kw_o)
    tag = (case, name, w, h, rp, mode)
    np.testing.assert_array_equal(o["image"], ref["image"], err_msg=str(tag))          # (NaN == NaN here)
    np.testing.assert_array_equal(o["grads"], ref["grads"], err_msg=str(tag))
    assert o["stats"]["segments"] == ref["stats"]["segments"], tag
    if mode == "gradient image":
        np.testing.assert_array_equal(o["grad_image"], ref["grad_image"], err_msg=str(tag))
    n_nan += int(not np.isfinite(ref["grads"]).all())
    if case % 100 == 0:
        print(f"{case:5d} {name:22s} {w:3d}x{h:<3d} spp {rp.spp} b{b} p{p:g} {mode:14s} rays {o['stats']['segments']:7d} deepest {o['stats']['deepest']:3d}  bit for bit", flush=True)
print(f"FUZZ ORACLE OK: {n_cases} renders, the restatement equals the reference bit for bit in image, gradients and ray counts "
      f"({n_nan} of them with the reference's own NaN) in {time.time() - t0:.0f} s")
