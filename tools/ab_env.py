#!/usr/bin/env python3
"""One workload, several ENVIRONMENT settings of the library, one fresh process each (the library reads its tuning
environment once): per-kernel HIP-event time, median over rounds.
Usage: tools/ab_env.py [--scene S] [--lib L] "VAR=1 VAR2=3" "VAR=2" ...   ("" = no setting)"""
import os, subprocess, sys
if os.environ.get("AB_ENV_CHILD"):
    import numpy as np
    sys.path.insert(0, '.')
    import __graft_entry__ as e
    pkg = e.load_package()
    scene = pkg.scene_by_name(os.environ.get("AB_SCENE", "mesh160x160")); cam = pkg.cornell_camera(512, 512)
    rp = pkg.RenderParams(spp=int(os.environ.get("AB_SPP", "64")), min_bounces=8, absorb=1.0, seed=1, batch_paths=int(os.environ.get("AB_BATCH", "0")))
    r = pkg.HipRenderer(0, lib_path=os.path.abspath(os.environ["AB_LIB"]) if os.environ.get("AB_LIB") else None)
    r.upload_scene(scene)
    UNB, F64 = bool(int(os.environ.get("AB_UNBIASED", "0"))), bool(int(os.environ.get("AB_F64", "0")))
    for _ in range(3):
        r.render(cam, rp, backward=True, unbiased=UNB, f64=F64)
    res = []
    for _ in range(7):
        img, g, st = r.render(cam, rp, backward=True, timing=True, unbiased=UNB, f64=F64)
        res.append([st["kernels"][k]["ms"] for k in pkg.KERNEL_NAMES] + [st["ms_total"]])
    m = np.median(np.array(res), 0)
    print(os.environ.get("AB_LABEL", "").ljust(52), " ".join(f"{v:9.3f}" for v in m), f" grad0 {g[0][0]:.6g}",
          "sha", __import__("hashlib").sha1(np.ascontiguousarray(img).tobytes() + np.ascontiguousarray(g).tobytes()).hexdigest()[:10],
          "segments", st.get("segments"))
    sys.exit(0)
args = sys.argv[1:]
scene, lib = "mesh160x160", ""
while args and args[0].startswith("--"):
    if args[0] == "--scene": scene = args[1]
    if args[0] == "--lib": lib = args[1]
    args = args[2:]
print("setting".ljust(52), " ".join(k[:9].rjust(9) for k in ["raygen", "intersect", "shade", "film", "backward", "gradreduc", "walk", "path"]), "host_ms".rjust(9))
for setting in args or [""]:
    env = dict(os.environ, AB_ENV_CHILD="1", AB_SCENE=scene, AB_LIB=lib, AB_LABEL=(os.path.basename(lib) + " " if lib else "") + (setting or "(default)"))
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
