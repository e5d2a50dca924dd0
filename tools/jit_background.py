#!/usr/bin/env python3
"""DRT_SPECIALISE_AUTO in a render loop: the frame times around the moment the scene's own kernel arrives (the compile runs
on a thread of the library's own; no frame waits for it)."""
import sys, time
sys.path.insert(0, ".")
import __graft_entry__ as e
pkg = e.load_package()
r = pkg.HipRenderer(0)
r.set_specialisation(pkg.SPECIALISE_AUTO)
r.upload_scene(pkg.scene_by_name(sys.argv[1] if len(sys.argv) > 1 else "random11"))
cam = pkg.cornell_camera(512, 512)
rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1)
t0 = time.perf_counter()
times = []
for i in range(2000):
    t1 = time.perf_counter()
    _, _, st = r.render(cam, rp, backward=True)
    times.append((time.perf_counter() - t1) * 1e3)
    if st["path_program"] == "specialised":
        print(f"frames 0..{i - 1} on the run-time program: first {times[0]:.2f} ms, median {sorted(times[:-1])[len(times) // 2]:.2f} ms, "
              f"slowest after the first {max(times[1:-1]):.2f} ms")
        print(f"frame {i}: the specialised kernel arrived {(time.perf_counter() - t0) * 1e3:.0f} ms after the first frame began "
              f"(this frame, which loaded it: {times[-1]:.2f} ms; jit_ms {st['jit_ms']:.0f})")
        nxt = []
        for _ in range(20):
            t1 = time.perf_counter(); r.render(cam, rp, backward=True); nxt.append((time.perf_counter() - t1) * 1e3)
        print(f"the next 20 frames: median {sorted(nxt)[10]:.2f} ms, slowest {max(nxt):.2f} ms")
        break
r.close()
