#!/usr/bin/env python3
"""Round-by-round comparison of every unbiased chain of ONE image row between the device's wavefront (f64) and the oracle:
prints the first round of every path whose suffix differs (number of suffix vertices, draws consumed).
Usage: DRT_HIP_DUMP_PATH=-2 DRT_ORACLE_TRACE_PATH=-2 tools/diag_chains.py scene W H spp min_bounces absorb seed row 2> log; then it parses log itself."""
import dataclasses, os, re, subprocess, sys
if os.environ.get("DRT_HIP_DUMP_PATH") != "-2":
    env = dict(os.environ, DRT_HIP_DUMP_PATH="-2", DRT_ORACLE_TRACE_PATH="-2")
    out = subprocess.run([sys.executable] + sys.argv, env=env, capture_output=True, text=True)
    name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    row = int(sys.argv[8])
    dev, orc = {}, {}
    for l in out.stderr.split("\n"):
        m = re.match(r"\[drt_hip\] path \d+ \(pixel (\d+) sample (\d+)\) round (\d+): chain prim (-?\d+), suffix base (\d+), (-?\d+) suffix vertices, draws after (\d+)", l)
        if m:
            px, sm, r, prim, base, n, after = map(int, m.groups())
            dev.setdefault(((row * w + px) * spp + sm), {})[r] = (base, n, after)
        m = re.match(r"\[oracle\] path (\d+) round (\d+): theta draw \d+, suffix base (\d+), (\d+) suffix vertices, draws after (\d+)", l)
        if m:
            p, r, base, n, after = map(int, m.groups())
            orc.setdefault(p, {})[r] = (base, n, after)
    print(out.stdout.strip())
    print(len(dev), "device chains,", len(orc), "oracle chains")
    for p in sorted(orc):
        for r in sorted(orc[p]):
            d = dev.get(p, {}).get(r)
            # (the oracle's base is the index of the suffix's first roulette draw; the device's is that of its first theta: one more when that depth draws)
            if d is None or d[1] != orc[p][r][1]:
                print(f"path {p} (pixel {p // spp % w}, sample {p % spp}) round {r}: oracle base {orc[p][r][0]} vertices {orc[p][r][1]} draws after {orc[p][r][2]}   device {d}")
                break
    sys.exit(0)
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package(); oracle = e.load_oracle()
name, w, h, spp, b, p, seed, row = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
scene = pkg.scene_by_name(name)
cam = pkg.cornell_camera(w, h) if not name.startswith("random") else pkg.Camera(w, h).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=seed, shard=row, n_shards=h, band_rows=1)
r = pkg.HipRenderer(0); r.upload_scene(scene)
o = oracle.render(scene, cam, rp, backward=True, unbiased=True, zero_dir_miss=True)
_, g, st = r.render(cam, rp, backward=True, f64=True, unbiased=True)
print("row", row, "oracle", o["stats"]["segments"], "device", st["segments"])
