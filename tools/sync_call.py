#!/usr/bin/env python3
"""What the SYNCHRONOUS call costs (drt_hip_render with host buffers: the call the reference's user makes, src/render.cpp:72-90
inside a gradient-descent loop): wall time per call against the kernels' own time, for small frames and for config 3's --
plain, without statistics, and with the image buffer pinned (drt_hip_pin_host).  One process per setting of the knobs:
  python tools/sync_call.py            all variants (subprocesses)
  python tools/sync_call.py one        this process, the environment as it is"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import __graft_entry__ as e
    pkg = e.load_package()
    r = pkg.HipRenderer(0)
    r.upload_scene(pkg.cornell_box())
    r.set_specialisation(pkg.SPECIALISE_NOW)
    print(f"{'frame':>16} {'mode':>26} {'call_us':>9} {'kernels_us':>11}")
    for size, spp, depth in ((64, 4, 4), (128, 16, 4), (256, 8, 4), (512, 64, 8)):
        cam = pkg.cornell_camera(size, size)
        rp = pkg.RenderParams(spp=spp, min_bounces=depth, absorb=1.0, seed=3)
        ref, gref, st = r.render(cam, rp, backward=True, timing=True)
        ker = sum(v["ms"] for v in st["kernels"].values()) * 1e3
        pinned = np.zeros((size, size, 3), dtype=np.float32)
        r.pin_host(pinned)
        plain = np.zeros((size, size, 3), dtype=np.float32)
        for mode, kw in (("fwd+bwd, stats", dict(want_stats=True)), ("fwd+bwd", dict(want_stats=False, img_out=plain)),
                         ("fwd+bwd, pinned image", dict(want_stats=False, img_out=pinned))):
            for _ in range(30):
                img, g, _ = r.render(cam, rp, backward=True, **kw)
            assert np.array_equal(img, ref) and np.array_equal(g, gref), mode
            n = 400 if size < 512 else 100
            t0 = time.perf_counter()
            for _ in range(n):
                r.render(cam, rp, backward=True, **kw)
            call = (time.perf_counter() - t0) / n * 1e6
            print(f"{size:>5}x{size:<5}x{spp:<3} {mode:>26} {call:9.1f} {ker:11.1f}", flush=True)
        r.unpin_host(pinned)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for name, env in (("round 4's way: hipMemcpyAsync + hipStreamSynchronize", {"DRT_HIP_SYNC_ZERO_COPY": "0", "DRT_HIP_SYNC_SPIN_US": "0"}),
                          ("image stored by the finishing kernel, hipStreamSynchronize", {"DRT_HIP_SYNC_SPIN_US": "0"}),
                          ("hipMemcpyAsync, completion word polled", {"DRT_HIP_SYNC_ZERO_COPY": "0"}),
                          ("default: image stored by the finishing kernel, completion word polled", {})):
            print("==", name, env, flush=True)
            out = subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
            print(out.stdout + out.stderr[-2000:], flush=True)
