#!/usr/bin/env python3
"""An inverse-rendering loop over the path (the use the reference is written for, /root/reference/README.md:88-101:
render -> loss -> backward -> parameter step), on the device through the C ABI:

    target  = render(scene with red = (0.5, 0, 0))                               once, 256 spp
    repeat:   image = drt_hip_render(seed A)                                     forward only
              adjoint = d loss / d pixel = 2 (image - target) / N                loss = mean squared error over the N pixel values
              gradient = drt_hip_render(seed B, BACKWARD, adjoint_rgb) / spp     out_param_grad is the SUM over the samples
              Adam step on the red albedo, clip to [0, 1], drt_hip_update_params

The image the adjoint comes from and the samples the gradient is taken on are INDEPENDENT (two seeds per step): with one
sample set for both, E[(I - T) dI] carries the covariance of a pixel's estimate with its own derivative, and the minimum of
the noisy objective sits ~9 % below the true albedo at 16 spp.  The per-pixel adjoint is exact for a loss on the pixel MEANS
(the ABI's adjoint_rgb seeds every sample of a pixel alike); the reference's per-sample `loss_func(radiance).backward()` is
the same thing for a loss that is linear in the radiance.

    python tools/fit_albedo.py [--size 128] [--spp 16] [--steps 60] [--async] [--oracle]

--async: the same loop through drt_hip_render_async / drt_hip_wait (frame i + 1 needs the parameters of step i, so frames
cannot overlap: what is measured is the call overhead).  --oracle: the CPU restatement instead of the device (checker;
used to choose the optimiser's constants in the build container, which has no GPU)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def fit(render, n_params, p_index, start, steps, spp, n_values, lr=0.08, decay=0.96, log=None):
    """render(params [P,3], seed, backward, adjoint) -> (image [H,W,3], grads [P,3] | None).  Adam on parameter p_index.
    -> (fitted rgb, history of rgb per step)"""
    params = render.params0.copy()
    params[p_index] = start
    m = np.zeros(3); v = np.zeros(3)
    b1, b2, eps = 0.8, 0.99, 1e-8
    hist = []
    for k in range(steps):
        img, _ = render(params, 1000 + 2 * k, False, None)
        adj = (2.0 * (img.astype(np.float64) - render.target) / n_values).astype(np.float32)
        _, grads = render(params, 1001 + 2 * k, True, adj)
        g = grads[p_index] / spp
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        step = lr * decay ** k * (m / (1 - b1 ** (k + 1))) / (np.sqrt(v / (1 - b2 ** (k + 1))) + eps)
        params[p_index] = np.clip(params[p_index] - step, 0.0, 1.0)
        hist.append(params[p_index].copy())
        if log:
            loss = float(((img.astype(np.float64) - render.target) ** 2).mean())
            log(f"step {k:3d}  loss {loss:.6f}  red = ({params[p_index][0]:.4f}, {params[p_index][1]:.4f}, {params[p_index][2]:.4f})")
    return params[p_index].copy(), hist


class DeviceRender:
    """The device through the C ABI (drt_hip_render, or drt_hip_render_async + drt_hip_wait)."""

    def __init__(self, pkg, size, spp, depth, use_async=False):
        self.pkg = pkg
        self.scene = pkg.cornell_box()
        self.cam = pkg.cornell_camera(size, size)
        self.spp, self.depth, self.use_async = spp, depth, use_async
        self.r = pkg.HipRenderer(0)
        self.r.upload_scene(self.scene)
        self.params0 = np.array(self.scene.params, dtype=np.float64)
        # the loop renders into the same two image buffers (the forward frame's, the gradient frame's), pinned once: the
        # finishing kernel stores a frame straight into its buffer
        self.img = [np.zeros((size, size, 3), dtype=np.float32) for _ in range(2)]
        if not use_async:
            for im in self.img:
                self.r.pin_host(im)
        self.calls = 0
        self.target, _ = self(self.params0, 1, False, None, spp=256)
        self.target = self.target.astype(np.float64)
        self.calls = 0

    def __call__(self, params, seed, backward, adjoint, spp=None):
        self.r.update_params(params)
        rp = self.pkg.RenderParams(spp=spp or self.spp, min_bounces=self.depth, absorb=1.0, seed=seed)
        self.calls += 1
        if self.use_async:
            img, g, _ = self.r.wait(self.r.render_async(self.cam, rp, backward=backward, adjoint=adjoint), want_stats=False)
        else:
            img, g, _ = self.r.render(self.cam, rp, backward=backward, adjoint=adjoint, img_out=self.img[1 if backward else 0], want_stats=False)
        return img, g


class OracleRender:
    """TEST INFRASTRUCTURE: the same loop on the CPU restatement."""

    def __init__(self, pkg, oracle, size, spp, depth):
        self.pkg, self.oracle = pkg, oracle
        self.scene = pkg.cornell_box()
        self.cam = pkg.cornell_camera(size, size)
        self.spp, self.depth = spp, depth
        self.params0 = np.array(self.scene.params, dtype=np.float64)
        self.target, _ = self(self.params0, 1, False, None, spp=256)

    def __call__(self, params, seed, backward, adjoint, spp=None):
        self.scene.params = [tuple(p) for p in params]
        rp = self.pkg.RenderParams(spp=spp or self.spp, min_bounces=self.depth, absorb=1.0, seed=seed)
        o = self.oracle.render(self.scene, self.cam, rp, backward=backward, adjoint=adjoint)
        return o["image"], o["grads"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--async", dest="use_async", action="store_true")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    import __graft_entry__ as e
    pkg = e.load_package()
    if a.oracle:
        render = OracleRender(pkg, e.load_oracle(), a.size, a.spp, a.depth)
    else:
        render = DeviceRender(pkg, a.size, a.spp, a.depth, a.use_async)
    t0 = time.time()
    rgb, hist = fit(render, len(render.params0), 0, np.array([0.2, 0.2, 0.2]), a.steps, a.spp, a.size * a.size * 3,
                    log=None if a.quiet else print)
    dt = time.time() - t0
    err = np.abs(rgb - np.array([0.5, 0.0, 0.0])).max()
    print(f"fitted red = ({rgb[0]:.4f}, {rgb[1]:.4f}, {rgb[2]:.4f})  max error {err:.4f}  "
          f"{a.steps} steps, {2 * a.steps} renders in {dt:.2f} s ({1e3 * dt / (2 * a.steps):.2f} ms per render"
          f"{', drt_hip_render_async + drt_hip_wait' if a.use_async else ''})")
    return 0 if err <= 1e-2 else 1


if __name__ == "__main__":
    sys.exit(main())
