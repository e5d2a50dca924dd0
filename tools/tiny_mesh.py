#!/usr/bin/env python3
"""How conservative are the BVH walk's f32 box tests when the mesh is small against the room?  The mesh of the config-4
scene shrunk about its centre by 1 ... 1/256: f32 segment counts and gradients against the f64 mode (whose box tests
have 29 more bits), and the number of rays the f32 walk lost (hits that the f64 walk finds and f32 does not show up as
path flips: |segment difference| and the pixel count above the f32 tolerance)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
cam = pkg.cornell_camera(256, 256)
rp = pkg.RenderParams(spp=16, min_bounces=6, absorb=1.0, seed=3)
for shrink in (1, 4, 16, 64, 256):
    s = pkg.scene_by_name("mesh40x40")
    v, idx, fm = s.meshes[0]
    c = v.mean(0)
    s.meshes[0] = ((v - c) / shrink + c, idx, fm)
    r.upload_scene(s)
    i32, g32, s32 = r.render(cam, rp, backward=True)
    i64, g64, s64 = r.render(cam, rp, backward=True, f64=True)
    bad = (np.abs(i32.astype(np.float64) - i64).max(-1) > 2e-4 * np.abs(i64).max()).sum()
    print(f"mesh / {shrink:3d}: segments f32 {s32['segments']} f64 {s64['segments']} (diff {s32['segments'] - s64['segments']:+d}), "
          f"pixels off {bad} of {256 * 256}, grad rel {np.abs(g32 - g64).max() / np.abs(g64).max():.2e}")
