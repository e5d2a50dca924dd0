#!/usr/bin/env python3
"""Traversal statistics of k_intersect_mesh from a -DDRT_BVH_STATS build of the library (debug only):
node visits (LDS / memory), leaf visits and triangle tests per ray.  Usage: tools/bvh_stats.py build/lib_stats.so [scene]"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
lib_path = os.path.abspath(sys.argv[1])
scene = pkg.scene_by_name(sys.argv[2] if len(sys.argv) > 2 else "mesh160x160")
r = pkg.HipRenderer(0, lib_path=lib_path)
r.upload_scene(scene)
cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=8, min_bounces=8, absorb=1.0, seed=1)
out = (C.c_ulonglong * 16)()
r.lib.drt_hip_debug_bvh_stats(out)
_, _, st = r.render(cam, rp, backward=True)
r.lib.drt_hip_debug_bvh_stats(out)
rays, lds, mem, leaves, tris = out[0], out[1], out[2], out[3], out[4]
print(f"rays {rays}  (segments {st['segments']})")
print(f"per ray: node visits from LDS {lds / rays:.2f}, from memory {mem / rays:.2f}, leaf visits {leaves / rays:.2f}, triangle tests {tris / rays:.2f}")
print(f"wave level: interior iterations {out[5]}, outer iterations {out[6]}, refill events {out[7]}; lanes busy per interior iteration {(lds + mem) / max(1, out[5]):.1f} of 64, "
      f"leaf lanes per outer iteration {leaves / max(1, out[6]):.1f}")
print(f"visits whose entry distance lies beyond the hit found meanwhile (a stack that kept the distance would skip them): "
      f"nodes {out[8] / rays:.2f} per ray = {100 * out[8] / max(1, lds + mem):.1f} % of node visits, leaves {out[9] / rays:.2f} per ray = {100 * out[9] / max(1, leaves):.1f} % of leaf visits")
print("rays by deepest stack: " + "  ".join(f"{n}: {100 * out[10 + i] / rays:.1f}%" for i, n in enumerate(["<=4", "<=8", "<=12", "<=16", "<=24", ">24"])))
print(f"16-byte lane accesses per ray: nodes {4 * mem / rays:.1f} + triangles {3 * tris / rays:.1f} + ray/hit 3")

hist = (C.c_ulonglong * 24)()
if hasattr(r.lib, "drt_hip_debug_bvh_hist") and r.lib.drt_hip_debug_bvh_hist(hist) == 0:
    names = ["1", "2", "3-4", "5-8", "9-16", "17-32", "33-64", ">64"]
    tot = sum(hist[i] for i in range(8)); totv = sum(hist[16 + i] for i in range(8))
    print("rays by node visits:   " + "  ".join(f"{n}: {100 * hist[i] / tot:.1f}%" for i, n in enumerate(names)))
    print("their share of visits: " + "  ".join(f"{n}: {100 * hist[16 + i] / totv:.1f}%" for i, n in enumerate(names)))
    print("ending on a triangle:  " + "  ".join(f"{n}: {100 * hist[8 + i] / max(1, hist[i]):.0f}%" for i, n in enumerate(names)) + f"   (all rays: {100 * sum(hist[8 + i] for i in range(8)) / tot:.1f}%)")
