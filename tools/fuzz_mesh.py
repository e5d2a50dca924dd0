#!/usr/bin/env python3
"""Random triangle meshes (displaced spheres of random resolution, with and without per-face albedo parameters) in the Cornell
box, random small frames, both integration operators: the device's f64 mode against the CPU restatement of the reference (its
raycast is the reference's linear scan over every triangle: no BVH to share a bug with) -- identical ray counts, gradients to
1e-9, image to f32 rounding; the f32 mode must stay finite and close.
Usage: tools/fuzz_mesh.py [n_cases] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import __graft_entry__ as e

pkg = e.load_package()
oracle = e.load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
r = pkg.HipRenderer(0)
worst64 = worst32 = 0.0
t0 = time.time()
for case in range(n_cases):
    lat, lon = int(rs.randint(3, 49)), int(rs.randint(3, 49))
    pf = int(rs.choice([0, 0, 3, 5, -1]))          # (-1: an albedo parameter of its own per face, drt_mesh_desc::face_param)
    scene = pkg.cornell_with_mesh(lat, lon, pf, seed=int(rs.randint(1 << 20)))
    w, h = int(rs.randint(6, 49)), int(rs.randint(6, 41))
    cam = pkg.cornell_camera(w, h)
    fixed = rs.rand() < 0.5
    b = int(rs.randint(1, 7))
    p = 1.0 if fixed else float(rs.choice([0.35, 0.5, 0.8]))
    unbiased = rs.rand() < 0.3
    rp = pkg.RenderParams(spp=int(rs.randint(1, 3 if unbiased else 5)), min_bounces=b, absorb=p, seed=int(rs.randint(1 << 30)),
                          batch_paths=int(rs.choice([0, 0, 257, 1500])),
                          bounces_per_launch=int(rs.choice([0, 0, 1])))     # (1: the queue wavefront where k_path_mesh would run)
    adjoint = rs.uniform(0.2, 1.5, (h, w, 3)).astype(np.float32) if rs.rand() < 0.3 else None
    o = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint, unbiased=unbiased, zero_dir_miss=unbiased)
    r.upload_scene(scene)
    img, g, st = r.render(cam, rp, backward=True, f64=True, unbiased=unbiased, adjoint=adjoint)
    img32, g32, st32 = r.render(cam, rp, backward=True, unbiased=unbiased, adjoint=adjoint)
    assert st["capped_paths"] == 0 and st["segments"] == o["stats"]["segments"], (case, lat, lon, pf, w, h, rp, unbiased, st["segments"], o["stats"]["segments"])
    scale = max(1e-300, float(np.abs(o["grads"]).max()))
    e64 = float(np.abs(g - o["grads"]).max() / scale)
    e32 = float(np.abs(g32 - o["grads"]).max() / scale)
    assert e64 < 1e-9, (case, lat, lon, pf, w, h, rp, unbiased, e64)
    np.testing.assert_allclose(img, o["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    assert np.isfinite(img32).all() and np.isfinite(g32).all() and abs(st32["segments"] - st["segments"]) <= 0.01 * st["segments"] + 8
    worst64, worst32 = max(worst64, e64), max(worst32, e32)
    print(f"{case:3d} mesh{lat}x{lon}{('f%d' % pf if pf > 0 else 'fall') if pf else '':4s} {2 * lon * (lat - 1):5d} triangles {w:3d}x{h:<3d} spp {rp.spp} b{b} p{p:g} {'unb' if unbiased else 'bia'} "
          f"{'adj' if adjoint is not None else '   '} batch {rp.batch_paths:4d} {'path ' if st['kernels']['path']['launches'] else 'queue'} rays {st['segments']:7d} (f32 {st32['segments'] - st['segments']:+d})  f64 mode {e64:.1e}  f32 mode {e32:.1e}", flush=True)
print(f"FUZZ MESH OK: {n_cases} cases in {time.time() - t0:.0f} s; worst gradient deviation from the restatement: f64 mode {worst64:.2e}, f32 mode {worst32:.2e}")
