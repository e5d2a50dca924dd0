import os, sys, json, subprocess
sys.path.insert(0, '.')
def one():
    import numpy as np
    import __graft_entry__ as e
    pkg = e.load_package()
    r = pkg.HipRenderer(0)
    r.upload_scene(pkg.scene_by_name("cornell"))
    cam = pkg.cornell_camera(512, 512)
    rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
    for _ in range(30):
        r.render(cam, rp, backward=True, want_stats=False)
    ts = []
    for _ in range(40):
        _, _, st = r.render(cam, rp, backward=True, timing=True)
        ts.append((st["kernels"]["path"]["ms"], st["kernels"]["film"]["ms"]))
    a = np.array(ts)
    print(json.dumps([float(np.median(a[:, 0])), float(np.min(a[:, 0])), float(np.median(a[:, 1]))]))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    one()
else:
    for rnd in range(2):
        for spr in (0, 4, 5, 6, 7, 8, 9, 10, 11, 13, 16, 22, 32):
            o = subprocess.run([sys.executable, __file__, "one"], env=dict(os.environ, DRT_HIP_PATH_SPR=str(spr)), capture_output=True, text=True, timeout=300)
            print(f"round {rnd} spr {spr:3d}  k_path median / min, finish: {o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-200:]}", flush=True)
