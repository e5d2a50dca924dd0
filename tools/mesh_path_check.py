"""k_path_mesh against the queue wavefront and the oracle, and their kernel times side by side.

  python tools/mesh_path_check.py parity            small frames: f64 vs the oracle (segments identical, 1e-9), f32 route vs route
  python tools/mesh_path_check.py time [W H SPP]    kernel milliseconds per frame, one subprocess per variant (the knobs are
                                                    read once per process)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def kernel_ms(st):
    return {k: round(v["ms"], 4) for k, v in st["kernels"].items() if v["ms"] > 0}


def parity():
    import numpy as np
    import __graft_entry__ as e
    pkg = e.load_package()
    oracle = e.load_oracle()
    r = pkg.HipRenderer(0)
    worst = 0.0
    cases = [("mesh10x12", 24, 24, 3, dict(min_bounces=4, absorb=1.0)),
             ("mesh40x40", 48, 48, 4, dict(min_bounces=5, absorb=1.0)),
             ("mesh10x12", 29, 28, 3, dict(min_bounces=1, absorb=0.5)),
             ("mesh40x40", 40, 30, 4, dict(min_bounces=2, absorb=0.3, max_depth=6))]
    for name, W, H, spp, kw in cases:
        scene = pkg.scene_by_name(name)
        cam = pkg.cornell_camera(W, H)
        rp = pkg.RenderParams(spp=spp, seed=3, **kw)
        r.upload_scene(scene)
        ref = oracle.render(scene, cam, rp, backward=True)
        gmax = np.abs(ref["grads"]).max()
        for f64 in (True, False):
            img, grads, st = r.render(cam, rp, backward=True, f64=f64)
            rq = pkg.RenderParams(spp=spp, seed=3, bounces_per_launch=1, **kw)
            imq, gq, stq = r.render(cam, rq, backward=True, f64=f64)
            assert st["kernels"]["path"]["launches"] == 1 and stq["kernels"]["path"]["launches"] == 0, (st["kernels"], stq["kernels"])
            ge = np.abs(grads - ref["grads"]).max() / gmax
            ie = np.abs(img - ref["image"]).max() / ref["image"].max()
            gr = np.abs(grads - gq).max() / gmax
            print(f"{name} {W}x{H}x{spp} {kw} f64={f64}: segments {st['segments']} / oracle {ref['stats']['segments']} / queue {stq['segments']}"
                  f"  capped {st['capped_paths']}/{stq['capped_paths']}  grad err {ge:.2e}  image err {ie:.2e}  route-vs-route grad {gr:.2e}"
                  f"  walked {st['kernels']['intersect_mesh']['units']} / {stq['kernels']['intersect_mesh']['units']}")
            if f64:
                assert st["segments"] == ref["stats"]["segments"] == stq["segments"]
                assert ge < 1e-9 and ie < 1e-6, (ge, ie)
                worst = max(worst, ge)
            else:
                assert ge < 2e-4, ge
    print("parity ok, worst f64 gradient deviation", worst)


def time_one(W, H, spp):
    import __graft_entry__ as e
    pkg = e.load_package()
    r = pkg.HipRenderer(0, lib_path=os.environ.get("CHECK_LIB"))
    r.upload_scene(pkg.scene_by_name("mesh160x160"))
    cam = pkg.cornell_camera(W, H)
    rp = pkg.RenderParams(spp=spp, min_bounces=8, absorb=1.0, seed=1)
    for _ in range(3):
        r.render(cam, rp, backward=True)
    best = None
    for _ in range(5):
        _, _, st = r.render(cam, rp, backward=True, timing=True)
        ms = sum(v["ms"] for v in st["kernels"].values())
        if best is None or ms < best[0]:
            best = (ms, st)
    ms, st = best
    print(json.dumps({"kernels_ms": round(ms, 3), "segments": st["segments"], "gray_s": round(st["segments"] / ms / 1e6, 2),
                      "per_kernel": kernel_ms(st), "walked": st["kernels"]["intersect_mesh"]["units"]}))


def small():
    """wall time per synchronous call, both routes, small frames (the frames of an optimisation loop)"""
    import time
    import __graft_entry__ as e
    pkg = e.load_package()
    r = pkg.HipRenderer(0)
    r.upload_scene(pkg.scene_by_name("mesh160x160"))
    print(f"{'frame':>14} {'route':>12} {'call_us':>9} {'kernels_us':>11} {'launches':>9} {'Mray/s':>9}")
    for W, spp, depth in ((64, 4, 4), (128, 16, 4), (128, 16, 8), (256, 8, 8), (256, 32, 8), (512, 8, 8), (512, 16, 8), (512, 64, 8)):
        cam = pkg.cornell_camera(W, W)
        for route, bpl in (("k_path_mesh", 0), ("queue", 1)):
            if route == "k_path_mesh" and os.environ.get("DRT_HIP_MESH_PATH_MAX") is None and W * W * spp > (1 << 20):
                continue
            rp = pkg.RenderParams(spp=spp, min_bounces=depth, absorb=1.0, seed=3, bounces_per_launch=bpl)
            for _ in range(10):
                r.render(cam, rp, backward=True)
            n = 100 if W * W * spp < 3e6 else 20
            t0 = time.perf_counter()
            for _ in range(n):
                r.render(cam, rp, backward=True)
            call = (time.perf_counter() - t0) / n * 1e6
            _, _, st = r.render(cam, rp, backward=True, timing=True)
            ker = sum(v["ms"] for v in st["kernels"].values()) * 1e3
            nl = sum(v["launches"] for v in st["kernels"].values())
            print(f"{W:>5}x{W:<4}x{spp:<3} {route:>12} {call:9.1f} {ker:11.1f} {nl:9d} {st['segments'] / call:9.1f}", flush=True)


def time_all(W, H, spp):
    big = {"DRT_HIP_MESH_PATH_MAX": str(1 << 30)}
    variants = [("queue", {"DRT_HIP_MESH_PATH_MAX": "0"}),
                ("k_path_mesh shade_min 32", {}),
                ("k_path_mesh shade_min 16", {"DRT_HIP_MESH_SHADE_MIN": "16"}),
                ("k_path_mesh shade_min 24", {"DRT_HIP_MESH_SHADE_MIN": "24"}),
                ("k_path_mesh shade_min 40", {"DRT_HIP_MESH_SHADE_MIN": "40"}),
                ("k_path_mesh shade_min 48", {"DRT_HIP_MESH_SHADE_MIN": "48"}),
                ("k_path_mesh descend_min 24", {"DRT_HIP_BVH_DESCEND_MIN": "24"}),
                ("k_path_mesh descend_min 40", {"DRT_HIP_BVH_DESCEND_MIN": "40"}),
                ("k_path_mesh spr 4", {"DRT_HIP_PATH_SPR": "4"}),
                ("k_path_mesh spr 16", {"DRT_HIP_PATH_SPR": "16"}),
                ("k_path_mesh spr 32", {"DRT_HIP_PATH_SPR": "32"}),
                ("k_path_mesh regen_min 4", {"DRT_HIP_PATH_REGEN_MIN": "4"}),
                ("k_path_mesh regen_min 16", {"DRT_HIP_PATH_REGEN_MIN": "16"})]
    extra = os.environ.get("CHECK_VARIANTS")
    if extra:
        big = {}
        variants = [(v, dict(kv.split("=") for kv in v.split(",") if kv)) for v in extra.split(";")]
    for name, env in variants:
        out = subprocess.run([sys.executable, __file__, "time_one", str(W), str(H), str(spp)], env=dict(os.environ, **env),
                             capture_output=True, text=True, timeout=600)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
        print(f"{name:32s} {line}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "parity"
    dims = [int(x) for x in sys.argv[2:5]] if len(sys.argv) >= 5 else [512, 512, 64]
    if mode == "parity":
        parity()
    elif mode == "small":
        small()
    elif mode == "time_one":
        time_one(*dims)
    else:
        time_all(*dims)
