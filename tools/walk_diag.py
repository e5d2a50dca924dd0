#!/usr/bin/env python3
"""Where does the BVH walk's time go?  (1) per depth: candidate rays and walk time of that depth's launch (renders with
depth caps 1..8, differences) -- the camera rays of depth 0 are coherent, the later ones are not; (2) with a
-DDRT_WALK_TIMES build: per wave of the last walk launch, when it started, when the list counters ran dry and when it
left -- how much of the launch is a tail of few waves.
(3) with a -DDRT_BVH_STATS build: the traversal statistics per depth.
Usage: tools/walk_diag.py [build/lib_times.so|-] [scene] [spp] [build/lib_stats.so]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
lib_times = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "-" else None
scene = pkg.scene_by_name(sys.argv[2] if len(sys.argv) > 2 else "mesh160x160")
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cam = pkg.cornell_camera(512, 512)

r = pkg.HipRenderer(0)
r.upload_scene(scene)
prev = (0.0, 0, 0.0, 0.0)
print("depth  candidates   walk_ms   ns/cand   shade_ms  segments_at_depth")
seg_prev = 0
for d in range(1, 9):
    rp = pkg.RenderParams(spp=spp, min_bounces=d, absorb=1.0, seed=1)
    for _ in range(2):
        r.render(cam, rp, backward=True)
    ms, sh = [], []
    for _ in range(5):
        _, _, st = r.render(cam, rp, backward=True, timing=True)
        ms.append(st["kernels"]["intersect_mesh"]["ms"]); sh.append(st["kernels"]["shade"]["ms"])
    m, s_ = float(np.median(ms)), float(np.median(sh))
    u = st["kernels"]["intersect_mesh"]["units"]
    dm, du = m - prev[0], u - prev[1]
    print(f"{d - 1:5d} {du:11d} {dm:9.3f} {1e6 * dm / max(1, du):9.3f} {s_ - prev[2]:9.3f}  {st['segments'] - seg_prev}")
    prev = (m, u, s_, 0.0)
    seg_prev = st["segments"]
r.close()

if lib_times:
    r = pkg.HipRenderer(0, lib_path=lib_times)
    r.upload_scene(scene)
    r.lib.drt_hip_debug_walk_times.argtypes = [C.c_void_p, C.c_int]
    print("depth  waves  span_us  mean_busy  busy/span  dry_at(p50)  end p50  p90  p99  max   [us from the first wave's start]")
    for d in range(1, 9):
        rp = pkg.RenderParams(spp=spp, min_bounces=d, absorb=1.0, seed=1)
        for _ in range(2):
            r.render(cam, rp, backward=True)
        buf = np.zeros((8192, 3), dtype=np.uint64)
        r.lib.drt_hip_debug_walk_times(buf.ctypes.data_as(C.c_void_p), 8192)
        t = buf[buf[:, 2] > 0].astype(np.int64)
        if len(t) == 0:
            continue
        t0 = t[:, 0].min()
        start, dry, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
        span = end.max()
        busy = (end - start).mean()
        dryv = dry[t[:, 1] > 0]
        print(f"{d - 1:5d} {len(t):6d} {span:8.1f} {busy:9.1f} {busy / span:9.3f} {np.median(dryv) if len(dryv) else -1:11.1f} "
              f"{np.percentile(end, 50):8.1f} {np.percentile(end, 90):6.1f} {np.percentile(end, 99):6.1f} {end.max():6.1f}   start p99 {np.percentile(start, 99):.1f}")
    r.close()

lib_stats = os.path.abspath(sys.argv[4]) if len(sys.argv) > 4 else None
if lib_stats:
    r = pkg.HipRenderer(0, lib_path=lib_stats)
    r.upload_scene(scene)
    out = (C.c_ulonglong * 16)()
    prev = np.zeros(8)
    print("depth      rays  nodes/ray(LDS+mem)  leaves/ray  tris/ray  lanes/interior-iter  leaf-lanes/outer-iter  iters per 64 rays (interior, outer)")
    for d in range(1, 9):
        rp = pkg.RenderParams(spp=min(spp, 16), min_bounces=d, absorb=1.0, seed=1)
        r.lib.drt_hip_debug_bvh_stats(out)
        r.render(cam, rp, backward=True)
        r.lib.drt_hip_debug_bvh_stats(out)
        cur = np.array([float(v) for v in out])
        dlt = cur - prev
        prev = cur
        rays = max(1.0, dlt[0])
        print(f"{d - 1:5d} {int(dlt[0]):9d}  {dlt[1] / rays:6.2f} + {dlt[2] / rays:5.2f}      {dlt[3] / rays:6.2f}    {dlt[4] / rays:6.2f}   "
              f"{(dlt[1] + dlt[2]) / max(1.0, dlt[5]):8.1f}              {dlt[3] / max(1.0, dlt[6]):8.1f}            {64 * dlt[5] / rays:6.1f} {64 * dlt[6] / rays:6.1f}")
    r.close()
