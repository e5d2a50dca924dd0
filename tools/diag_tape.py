#!/usr/bin/env python3
"""Dump the tapes (f32 and f64 mode) of every sample of one pixel: which sample differs, and where.
Usage: tools/diag_tape.py <scene> <w> <h> <spp> <min_bounces> <absorb> <seed> <x> <y>"""
import os, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
name, w, h, spp, b, p, seed, x, y = sys.argv[1], *map(int, sys.argv[2:6]), float(sys.argv[6]), *map(int, sys.argv[7:10])
sc = pkg.scene_by_name(name)
print("materials", sc.materials)
print("shapes", [(t, m, em) for (t, m, em, _) in sc.shapes])
r.upload_scene(sc)
cam = pkg.cornell_camera(w, h)
rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=seed)
for s in range(spp):
    i = s * w * h + y * w + x          # sample-major batch-local index (one batch)
    os.environ["DRT_HIP_DUMP_PATH"] = str(i)
    print(f"--- sample {s} (path {i}) f32", flush=True)
    r.render(cam, rp, backward=True)
    sys.stderr.flush()
    print(f"--- sample {s} f64", flush=True)
    r.render(cam, rp, backward=True, f64=True)
    sys.stderr.flush()
