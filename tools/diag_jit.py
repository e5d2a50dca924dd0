#!/usr/bin/env python3
"""Where do the specialised program's bits differ from the run-time program's?  (image / gradient differences per scene)"""
import os, subprocess, sys
import numpy as np
if os.environ.get("DJ_CHILD"):
    sys.path.insert(0, '.')
    import __graft_entry__ as e
    pkg = e.load_package()
    name = os.environ["DJ_SCENE"]
    scene = pkg.cornell_box() if name == "cornell" else (pkg.random_scene(int(name[4:]), specular=False) if name.startswith("diff") else pkg.scene_by_name(name))
    cam = pkg.cornell_camera(96, 64); rp = pkg.RenderParams(spp=6, min_bounces=5, absorb=1.0, seed=3)
    r = pkg.HipRenderer(0); r.set_specialisation(int(os.environ["DJ_MODE"])); r.upload_scene(scene)
    img, g, st = r.render(cam, rp, backward=bool(int(os.environ.get("DJ_BWD", "1"))))
    np.save(os.environ["DJ_OUT"] + ".img.npy", img)
    if g is not None: np.save(os.environ["DJ_OUT"] + ".g.npy", g)
    print(name, os.environ["DJ_MODE"], st["path_program"], st["segments"], st["jit_ms"])
    sys.exit(0)
for name, env in [("cornell", {"DRT_HIP_BUILTIN_PROGRAM": "0"}), ("cornell", {}), ("diff3", {}), ("random3", {}), ("cornell_specular", {"DRT_HIP_BUILTIN_PROGRAM": "0"})]:
    for bwd in ("1", "0"):
        outs = []
        for mode in ("-1", "2"):
            out = f"/tmp/dj_{name}_{mode}"
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, DJ_CHILD="1", DJ_SCENE=name, DJ_MODE=mode, DJ_OUT=out, DJ_BWD=bwd, **env), check=True)
            outs.append(out)
        a, b = np.load(outs[0] + ".img.npy"), np.load(outs[1] + ".img.npy")
        d = np.abs(a.astype(np.float64) - b)
        print(f"  {name} {env} bwd={bwd}: image differs in {int((d.max(-1) > 0).sum())} of {d.shape[0] * d.shape[1]} pixels, max abs {d.max():.3g} (max value {np.abs(a).max():.3g})")
        if bwd == "1":
            ga, gb = np.load(outs[0] + ".g.npy"), np.load(outs[1] + ".g.npy")
            print("     grads max rel diff", np.abs(ga - gb).max() / np.abs(ga).max())
