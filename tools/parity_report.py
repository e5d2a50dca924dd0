#!/usr/bin/env python3
"""Print the measured deviation of the f32 / f64 device modes from every golden fixture
(run on the GPU box). Used to set and to audit the tolerances written in tests/."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from conftest import SMALL_GOLDENS, UNBIASED_GOLDENS, case_inputs, load_golden  # noqa: E402

pkg = entry.load_package()
hip = pkg.HipRenderer(0)
names = SMALL_GOLDENS + UNBIASED_GOLDENS + ["c1_cornell_256x256x8_d4"] + (["c3_cornell_512x512x64_d8"] if "--big" in sys.argv else [])
print(f"{'fixture':34s} {'mode':4s} {'dseg':>6s} {'bad_px':>7s} {'max_px_err':>11s} {'mean_rel':>10s} {'grad_rel':>10s}")
for name in names:
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    for mode in ("f32", "f64"):
        img, grads, st = hip.render(cam, rp, backward=True, adjoint=adjoint, f64=(mode == "f64"),
                                    unbiased=name in UNBIASED_GOLDENS)
        im = img.astype(np.float64)
        if "image" in g:
            gi = g["image"].astype(np.float64)
            scale = np.abs(gi).max()
            err = np.abs(im - gi).max(-1)
            bad = int((err > 2e-4 * scale).sum())
            mx = err.max() / scale
        else:
            bad, mx = -1, float("nan")
        mean_rel = np.abs(im.mean((0, 1)) - g["mean_rgb"]).max() / g["mean_rgb"].max()
        grel = np.abs(grads - g["grads"]).max() / np.abs(g["grads"]).max()
        print(f"{name:34s} {mode:4s} {st['segments'] - int(g['segments']):6d} {bad:7d} {mx:11.3e} {mean_rel:10.3e} {grel:10.3e}")
hip.close()
