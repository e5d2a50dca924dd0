#!/usr/bin/env python3
"""The reference's default roulette (-b 1 -p 0.5, args.hpp:44-59) on config 3's frame: k_path<REGEN>'s launch time under the
knobs that shape its waves (one subprocess per setting).  python tools/roulette_sweep.py [one]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import __graft_entry__ as e
    pkg = e.load_package()
    r = pkg.HipRenderer(0, lib_path=os.environ.get("SWEEP_LIB"))
    r.upload_scene(pkg.cornell_box())
    r.set_specialisation(pkg.SPECIALISE_NOW)
    cam = pkg.cornell_camera(512, 512)
    # (SWEEP_MODE=d8: config 3 itself, every path to depth 8 -- the lockstep kernel -- under the same knobs)
    rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1) if os.environ.get("SWEEP_MODE") == "d8" else \
         pkg.RenderParams(spp=64, min_bounces=1, absorb=0.5, seed=1)
    for _ in range(5):
        r.render(cam, rp, backward=True)
    best = None
    for _ in range(15 if os.environ.get("SWEEP_MODE") == "d8" else 9):
        _, _, st = r.render(cam, rp, backward=True, timing=True)
        ms = st["kernels"]["path"]["ms"]
        best = ms if best is None else min(best, ms)
    print(json.dumps({"k_path_ms": round(best, 4), "segments": st["segments"], "gray_s": round(st["segments"] / best / 1e6, 1),
                      "program": st["path_program"]}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        variants = [""] + [f"DRT_HIP_PATH_SPR={s}" for s in (64, 32, 16, 8, 4)] + [f"DRT_HIP_PATH_REGEN_MIN={m}" for m in (1, 4, 16, 24, 32)] + \
                   ["DRT_HIP_PATH_REGEN=0"]
        if os.environ.get("SWEEP_ONLY_EXTRA"):
            variants = [""]
        variants += os.environ.get("SWEEP_EXTRA", "").split(";") if os.environ.get("SWEEP_EXTRA") else []
        for v in variants:
            env = dict(os.environ, **dict(kv.split("=") for kv in v.split(",") if kv))
            out = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True, timeout=600)
            print(f"{v or 'default':40s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
