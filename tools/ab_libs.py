#!/usr/bin/env python3
"""Interleaved A/B of several builds of libdrt_hip in ONE process (cdna guide rule 24): per-kernel
HIP-event time of the config-3 step, median over rounds.  Usage: tools/ab_libs.py lib1.so lib2.so ..."""
import sys, os
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
libs = sys.argv[1:]
scene = pkg.scene_by_name(os.environ.get("AB_SCENE", "cornell")); cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
UNB = bool(os.environ.get("AB_UNBIASED"))
F64 = bool(os.environ.get("AB_F64"))
rs = [pkg.HipRenderer(0, lib_path=os.path.abspath(l)) for l in libs]
for r in rs:
    r.upload_scene(scene)
    for _ in range(2):
        r.render(cam, rp, backward=True, unbiased=UNB, f64=F64)
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get('AB_ROUNDS', '7'))):
    for l, r in zip(libs, rs):
        _, _, st = r.render(cam, rp, backward=True, timing=True, unbiased=UNB, f64=F64)
        res[l].append([st["kernels"][k]["ms"] for k in pkg.KERNEL_NAMES] + [st["ms_total"]])
print("lib".ljust(44), " ".join(k[:9].rjust(9) for k in pkg.KERNEL_NAMES), "host_ms".rjust(9))
for l in libs:
    a = np.array(res[l])
    m = np.median(a, 0)
    tot = a[:, :-1].sum(1)
    print(os.path.basename(l).ljust(44), " ".join(f"{v:9.3f}" for v in m), f"  kernels: median {np.median(tot):.4f} mean {tot.mean():.4f} q25-q75 {np.percentile(tot, 25):.4f}-{np.percentile(tot, 75):.4f}")
