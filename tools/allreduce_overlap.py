#!/usr/bin/env python3
"""Frames per second of device-pointer renders with and without the in-library all-reduce (a 1-rank communicator on a
single-GPU box): does the collective of frame i keep frame i + 1's path kernel from overlapping?  (No.)  And the launch-order
effect of a torch.distributed process group in the same process:
    tools/allreduce_overlap.py                                          no process group
    tools/allreduce_overlap.py --torch-dist-eager                       group and its first collective, then the context: fast
    tools/allreduce_overlap.py --torch-dist-eager --context-first       context, then the group: fast
    tools/allreduce_overlap.py --torch-dist-eager --context-between     group, context, THEN the group's first collective: frames no longer overlap"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
import __graft_entry__ as e
pkg = e.load_package()
LATE = len(sys.argv) > 2 and sys.argv[2] == "--context-first"      # bench.py's order: the context exists before the process group's first collective
if LATE:
    r0 = pkg.HipRenderer(0); r0.upload_scene(pkg.cornell_box())
if len(sys.argv) > 1:                  # --torch-dist: with torch.distributed's own RCCL process group alive in the process (what bench.py --gpus N has)
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    if sys.argv[1] == "--torch-dist-eager":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl")
    BETWEEN = len(sys.argv) > 2 and sys.argv[2] == "--context-between"   # the slow order: group made, context made, THEN the group's first collective
    if BETWEEN:
        r0 = pkg.HipRenderer(0); r0.upload_scene(pkg.cornell_box()); LATE = True
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier(); torch.cuda.synchronize()
    print("torch.distributed initialised:", sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", flush=True)
r = r0 if LATE else pkg.HipRenderer(0)
r.upload_scene(pkg.cornell_box())
r.comm_init(pkg.comm_unique_id(), 0, 1)
cam = pkg.cornell_camera(512, 512)
dev = torch.device("cuda", 0)
out = torch.zeros((512, 512, 3), dtype=torch.float32, device=dev)
grads = [torch.zeros((4, 3), dtype=torch.float64, device=dev) for _ in range(2)]
for tag, flags in (("no collective", 0), ("DRT_RENDER_ALLREDUCE (same stream)", pkg.RENDER_ALLREDUCE), ("DRT_RENDER_ALLREDUCE_ASYNC (second stream)", pkg.RENDER_ALLREDUCE_ASYNC),
                   ("no collective", 0)):
    rp = pkg.RenderParams(spp=64, min_bounces=8, absorb=1.0, seed=1, flags=flags)
    for i in range(60):
        r.render_device(cam, rp, out.data_ptr(), grads[i & 1].data_ptr(), backward=True, sync=False)
    r.synchronize(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for i in range(100):
            r.render_device(cam, rp, out.data_ptr(), grads[i & 1].data_ptr(), backward=True, sync=False)
        r.synchronize(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    print(f"{tag:45s} {best:.4f} ms per frame", flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == "--barriers":
        dist.barrier()
