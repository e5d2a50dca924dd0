#!/bin/bash
set -u
O=gpurun_out/${1:-sab2}; mkdir -p $O
AB_ROUNDS=21 python3 tools/ab_libs.py build/lib_prev.so differentiable-renderer_amd/libdrt_hip.so > $O/ab.txt 2>&1
cat $O/ab.txt
python3 - > $O/bits.txt 2>&1 <<PY
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as e
pkg = e.load_package()
import os
for scene_name, b, p, md in (("cornell", 8, 1.0, 0), ("cornell_specular", 5, 1.0, 0), ("cornell", 2, 0.3, 6), ("random3", 4, 1.0, 0)):
    outs = []
    for lib in ("build/lib_prev.so", "differentiable-renderer_amd/libdrt_hip.so"):
        r = pkg.HipRenderer(0, lib_path=os.path.abspath(lib)); r.upload_scene(pkg.scene_by_name(scene_name))
        rp = pkg.RenderParams(spp=8, min_bounces=b, absorb=p, seed=4, **({"max_depth": md} if md else {}))
        outs.append(r.render(pkg.cornell_camera(160, 128), rp, backward=True)); r.close()
    print(scene_name, b, p, md, "bit-identical:", np.array_equal(outs[0][0], outs[1][0]), np.array_equal(outs[0][1], outs[1][1]), outs[0][2]["segments"] == outs[1][2]["segments"], outs[0][2]["capped_paths"], outs[1][2]["capped_paths"])
PY
cat $O/bits.txt
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
tail -4 $O/tests.txt
