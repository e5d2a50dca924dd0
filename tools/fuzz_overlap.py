#!/usr/bin/env python3
"""Random sequences of renders on ONE context -- device-pointer frames that do not wait (their k_path grids overlap on two
streams), synchronous host renders, parameter updates, scene uploads, frame sizes, both estimators, both precisions, an
adjoint now and then -- every frame compared bit for bit with a second context that only renders synchronously.
Usage: tools/fuzz_overlap.py [n_ops] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import __graft_entry__ as e
pkg = e.load_package()
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
n_ops = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
a, b = pkg.HipRenderer(0), pkg.HipRenderer(0)          # a: the context under test, b: the synchronous twin
names = ["cornell", "cornell_specular", "cornell_walls", "random3", "mesh6x8",
         "cornell_shapes", "params12", "cornell_disc_box", "cornell_coslobe"]      # (round 6: the general form, caller-defined kinds)
scene = None
pending = []                                            # (out tensor, grad tensor | None, expected image, expected grads | None)
def flush():
    a.synchronize()
    for o, g, ei, eg in pending:
        np.testing.assert_array_equal(o.cpu().numpy(), ei)
        if g is not None:
            np.testing.assert_array_equal(g.cpu().numpy(), eg)
    n = len(pending); pending.clear(); return n
checked = 0
for op in range(n_ops):
    k = rs.rand()
    if scene is None or k < 0.06:
        checked += flush()
        name = names[rs.randint(len(names))]
        scene = pkg.scene_by_name(name); params = np.array(scene.params)
        a.upload_scene(scene); b.upload_scene(scene)
        continue
    if k < 0.14:
        params = np.clip(params * rs.uniform(0.8, 1.2, params.shape), 0.05, 4.0)
        a.update_params(params); b.update_params(params)          # (blocks until the frames in flight are done)
        continue
    w, h = int(rs.choice([64, 96, 128])), int(rs.choice([48, 64]))
    cam = pkg.cornell_camera(w, h) if not name.startswith("random") else pkg.Camera(w, h).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
    fixed = rs.rand() < 0.6
    rp = pkg.RenderParams(spp=int(rs.randint(1, 9)), min_bounces=int(rs.randint(1, 7)), absorb=1.0 if fixed else float(rs.choice([0.3, 0.5])),
                          seed=int(rs.randint(1 << 30)))
    backward = rs.rand() < 0.8
    unb = backward and rs.rand() < 0.2
    f64 = rs.rand() < 0.15
    adj = rs.uniform(0.2, 1.5, (h, w, 3)).astype(np.float32) if backward and rs.rand() < 0.25 else None
    exp = b.render(cam, rp, backward=backward, unbiased=unb, f64=f64, adjoint=adj)
    if rs.rand() < 0.2:                                  # a synchronous render in between (host buffers)
        got = a.render(cam, rp, backward=backward, unbiased=unb, f64=f64, adjoint=adj)
        np.testing.assert_array_equal(got[0], exp[0])
        if backward: np.testing.assert_array_equal(got[1], exp[1])
        checked += 1
        continue
    o = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    g = torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) if backward else None
    d_adj = torch.from_numpy(adj).to(dev) if adj is not None else None
    if d_adj is not None: torch.cuda.synchronize()
    flags = (pkg.RENDER_F64 if f64 else 0) | (pkg.RENDER_UNBIASED if unb else 0)
    import dataclasses
    a.render_device(cam, dataclasses.replace(rp, flags=flags), o.data_ptr(), g.data_ptr() if g is not None else 0,
                    adjoint_ptr=d_adj.data_ptr() if d_adj is not None else 0, backward=backward)
    pending.append((o, g, exp[0], exp[1] if backward else None))
    if d_adj is not None or len(pending) >= 6 or rs.rand() < 0.15:
        checked += flush()
checked += flush()
print(f"FUZZ OVERLAP OK: {n_ops} operations, {checked} frames compared bit for bit with synchronous renders of a second context")
