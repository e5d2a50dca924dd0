#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Metric: Mray/s (fwd+bwd) on the Cornell box of /root/reference/src/render.cpp:26-59,
512x512, 64 spp, depth 8 (`-b 8 -p 1`), gradients w.r.t. the material albedos + emission
(BASELINE config 3).  A "step" is one full render call: every path of the frame traced from the
eye to its end, gradients accumulated and reduced.  ray = one raycast (camera ray included),
exactly what the reference's Pathtracer::raycast counts.

  python bench.py [--gpus N --steps K --warmup W] [--config 2|3|4|5]

--gpus N > 1 without torchrun's environment: this process starts N ranks itself (a fresh
`python -m torch.distributed.run` child, one rank per GPU) and relays rank 0's JSON line.  Under
torchrun (RANK set) it is one of the ranks.  The frame's rows are dealt to the ranks in
interleaved 16-row bands and spp is multiplied by N, so every rank traces the same number of
paths whatever N is (weak scaling); the only collective is ONE ncclAllReduce of the P x 3 f64
gradient vector per step, enqueued by libdrt_hip.so itself on the context's stream
(drt_hip_comm_init_rank + DRT_RENDER_ALLREDUCE).  Prints ONE JSON line (rank 0).

--config selects a BASELINE.json configuration; per GPU it is that configuration's 1/8 share
(configs 4 and 5 are quoted on 8 GPUs), so `--config 4 --gpus 8` is exactly config 4.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The path kernels of consecutive frames overlap on two streams of the render context, and HIP shares its hardware queues
# among ALL streams of the process: four by default.  A torch.distributed RCCL process group brings streams of its own, and
# with the context made between the group's creation and its first collective two of the context's streams ended up on one
# queue -- frames in stream order, 0.716 instead of 0.658 ms (tools/allreduce_overlap.py --torch-dist-eager --context-between:
# 0.716 with 2 or 4 queues, 0.657 with 8 or 16; every other workload of this file measures the same with 4 and 8).  Read by
# the HIP runtime when it starts, so it is set before torch is imported; a value from the environment wins.  One process per
# GPU, as the ranks of this file are: two processes of 8 queues each on ONE device oversubscribe its hardware queues and the
# scheduler time-slices them (--same-gpu, the plumbing test: 45.7 ms per step instead of 1.7) -- there the default stays.
if "--same-gpu" not in sys.argv:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6200.0        # measured float4 copy on the pool's boxes (profiles/r01_microbench_stream_roofs.txt; guide: 6.29)
# vector-instruction issue: a wave64 VALU op takes 2 cycles of its SIMD; 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0     # 1228.8 G wave-instructions / s
VALU_MEASURED_GINST = 900.0               # what v_fma_f32 alone sustains on the pool's boxes (profiles/r04_microbench_valu_issue.txt)

# BASELINE.json configs -> workload per GPU (configs 4, 5: the 1/8 share of one of the 8 GPUs they are quoted on)
CONFIGS = {
    2: dict(width=512, height=512, spp=64, depth=8, scene="cornell", forward_only=True,
            name="config 2: Cornell 512x512, 64 spp, depth 8, diffuse+emissive, fwd only"),
    3: dict(width=512, height=512, spp=64, depth=8, scene="cornell", forward_only=False,
            name="config 3: Cornell 512x512, 64 spp, depth 8, fwd + radiative backprop w.r.t. the albedos"),
    4: dict(width=1024, height=1024, spp=32, depth=8, scene="mesh160x160", forward_only=False,
            name="config 4: 1024x1024, 256 spp over 8 GPUs (32 spp per GPU), 50,880-triangle mesh in the box, fwd+bwd"),
    5: dict(width=2048, height=2048, spp=128, depth=16, scene="cornell_specular", forward_only=False,
            name="config 5: 2048x2048, 1024 spp over 8 GPUs (128 spp per GPU), depth 16, diffuse + specular, fwd+bwd"),
}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start N ranks as a CHILD process tree (never an exec of this process:
    under rocprofv3 the GPU is initialised before main() runs) and pass rank 0's line through."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def load_distributed():
    """differentiable-renderer_amd/distributed.py (the directory name has a hyphen)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("drt_distributed", os.path.join(ROOT, "differentiable-renderer_amd", "distributed.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _oracle_shard(args):
    """Worker of the all-cores CPU figure: one row-band shard of the sample, in its own process."""
    import __graft_entry__ as entry
    scene_name, w, h, spp, min_b, absorb, backward, shard, n_shards = args
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    rp = pkg.RenderParams(spp=spp, min_bounces=min_b, absorb=absorb, seed=1, shard=shard, n_shards=n_shards, band_rows=4)
    r = oracle.render(pkg.scene_by_name(scene_name), pkg.cornell_camera(w, h), rp, backward=backward)
    return r["stats"]["segments"]


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configuration (per-GPU share)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel PER GPU")
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--scene", default="", help="cornell | cornell_specular | mesh<lat>x<lon>[f<n>] | random<seed>")
    ap.add_argument("--absorb", type=float, default=1.0,
                    help="roulette absorption probability (the reference's -p; 1 = every path ends at --depth, the configs' setting)")
    ap.add_argument("--min-bounces", type=int, default=0,
                    help="the reference's -b; default = --depth.  '--absorb 0.5 --min-bounces 1' = the reference's own defaults")
    ap.add_argument("--per-face", action="store_true",
                    help="mesh scenes: an albedo parameter of its own for EVERY face (drt_mesh_desc::face_param; config 4 as SURVEY 8d "
                         "words it: 50,880 + 4 parameters, a 1.2 MB gradient vector through K6's fp64 atomics, K7 and the all-reduce)")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--unbiased", action="store_true", help="backward with the unbiased integration operator")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-views", action="store_true",
                    help="skip the extra measurements (one launch per bounce, f64, forward-only, host buffers): profiling "
                         "runs keep their kernel statistics unmixed")
    ap.add_argument("--bounces-per-launch", type=int, default=0, help="0 = automatic (drt_hip.h)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the real thing) | gloo: rendezvous only, lets ranks share ONE GPU to exercise the "
                         "N > 1 path on a single-GPU box (with --same-gpu); the numbers of such a run mean nothing")
    ap.add_argument("--same-gpu", action="store_true", help="every rank uses device 0 (testing only)")
    ap.add_argument("--single-process", action="store_true",
                    help="the ABI's OTHER multi-GPU form: ONE process, drt_hip_create_group over --gpus devices (host buffers, the "
                         "gradient summed inside the call by one grouped ncclAllReduce); an auxiliary measurement, not the contract line's mode")
    ap.add_argument("--store-n1", action="store_true",
                    help="1-GPU runs: store value under this workload's key in profiles/n1_reference.json (N > 1 runs print "
                         "weak_scaling.efficiency against it)")
    ap.add_argument("--reduce", default="library", choices=["library", "torch"],
                    help="where the gradient all-reduce runs: inside libdrt_hip.so (RCCL, default) or torch.distributed")
    ap.add_argument("--allreduce", default="async", choices=["async", "stream"],
                    help="library all-reduce on the context's second stream, overlapping the next step (async, default) or "
                         "in stream order between two steps (stream)")
    ap.add_argument("--preheat-ms", type=float, default=60.0,
                    help="setup: drive the step loop for this long before the warm-up steps (device clocks out of idle); 0 = none")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--ref-seconds", type=float, default=8.0,
                    help="target duration of the sample the reference itself (oracle/_ref/ref_harness, when present) is timed on; 0 = skip")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the oracle on every host core (independent row-band processes)")
    a = ap.parse_args(argv)
    cfg = CONFIGS[a.config]
    a.width = a.width or cfg["width"]
    a.height = a.height or cfg["height"]
    a.spp = a.spp or cfg["spp"]
    a.depth = a.depth or cfg["depth"]
    a.scene = a.scene or cfg["scene"]
    if a.per_face and a.scene.startswith("mesh") and "f" not in a.scene[4:]:
        a.scene += "fall"
    a.forward_only = a.forward_only or cfg["forward_only"]
    a.min_bounces = a.min_bounces or a.depth
    a.roulette = a.absorb < 1.0
    a.is_config = all(getattr(a, k) == v or (k == "scene" and a.per_face and getattr(a, k) == v + "fall") for k, v in cfg.items() if k != "name") and \
                  not a.roulette and a.min_bounces == a.depth
    a.config_name = (cfg["name"] + (", an albedo parameter per face" if a.per_face else "")) if a.is_config else "custom"
    return a


def single_process_main(a):
    """--single-process: one process, drt_hip_create_group over a.gpus devices.  Weak scaling like the contract's mode: every
    device renders a.spp samples of its interleaved row bands' pixels... of a frame of a.spp x N samples per pixel."""
    import json
    import time
    import numpy as np
    import __graft_entry__ as entry
    pkg = entry.load_package()
    n = max(1, a.gpus)
    have = pkg.load_library().drt_hip_device_count()
    if have < n and not a.same_gpu:
        print(f"bench.py --single-process: {n} devices asked for, {have} present", file=sys.stderr)
        return 3
    ids = [0] * n if a.same_gpu else list(range(n))
    r = pkg.HipRenderer(ids)
    r.set_specialisation(pkg.SPECIALISE_NOW)
    scene = pkg.scene_by_name(a.scene)
    r.upload_scene(scene)
    cam = pkg.cornell_camera(a.width, a.height)
    backward = not a.forward_only
    rp = pkg.RenderParams(spp=a.spp * n, min_bounces=a.min_bounces, absorb=a.absorb, seed=1, band_rows=16, batch_paths=a.batch_paths,
                          bounces_per_launch=a.bounces_per_launch)
    img = np.zeros((a.height, a.width, 3), dtype=np.float32)
    t_end = time.perf_counter() + a.preheat_ms * 1e-3
    while time.perf_counter() < t_end:
        r.render(cam, rp, backward=backward, img_out=img, want_stats=False)
    for _ in range(a.warmup):
        r.render(cam, rp, backward=backward, img_out=img, want_stats=False)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r.render(cam, rp, backward=backward, img_out=img, want_stats=False)
    dt = (time.perf_counter() - t0) / a.steps
    _, _, st = r.render(cam, rp, backward=backward, timing=True)
    devices = [r.pci_bus_id(i) for i in range(n)]
    line = {"metric": f"Mray/s ({'fwd+bwd' if backward else 'fwd'}), {a.scene} {a.width}x{a.height} @{a.spp * n}spp, one process, drt_hip_create_group",
            "value": round(st["segments"] / dt * 1e-6, 2), "unit": "Mray/s", "n_gpus": n, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{a.config_name}; scene '{a.scene}' {a.width}x{a.height}, {a.spp} spp per GPU",
                       "parallelism": f"1 process, {n} devices ({'; '.join(devices)}) in a group context (drt_hip_create_group): interleaved 16-row "
                                      f"bands, host buffers, the gradient summed inside the call (members of one device added on it, the "
                                      f"distinct devices by ONE grouped ncclAllReduce over {len(set(devices))} ranks)",
                       "mode": "synchronous host-buffer calls (PCIe D2H of every member's rows inside the step): an auxiliary view, "
                               "the contract line is `bench.py --gpus N` (one process per GPU, device buffers)"},
            "segments_per_step": st["segments"],
            "kernels_ms_slowest_member": {k: round(v["ms"], 4) for k, v in st["kernels"].items() if v["ms"] > 0},
            "roofline": None, "cpu_baseline": None}
    r.close()
    print(json.dumps(line), flush=True)
    return 0


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    if a.single_process:
        sys.exit(single_process_main(a))
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus, argv))

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, a.gpus) and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # launched by torch.distributed.run (also with one rank: exercises the in-library RCCL path on one GPU)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if use_dist and rank != 0:
        # the launcher merges every rank's stdout, and RCCL writes a version banner there through C stdio when a communicator
        # is made (it leaves the buffer when the process ends: possibly behind rank 0's line).  Only rank 0 has anything to say.
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if a.same_gpu:
        local_rank = 0
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # The process group runs its first collective NOW, before this rank's render context makes its streams.  Measured
            # (tools/allreduce_overlap.py, one rank): with the context created between init_process_group and the group's
            # first collective, the path kernels of consecutive frames never ran side by side -- 0.716 ms per frame instead
            # of 0.658, in-library all-reduce or not; with one collective first, 0.658 in every mode.  (Two of the context's
            # streams then share one of HIP's four hardware queues; GPU_MAX_HW_QUEUES=8, set at the top of this file, removes
            # the effect by itself -- both are kept.)  INTEGRATION.md section 3 says so.
            first = torch.ones(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(first)
            torch.cuda.synchronize()
        else:
            dist.init_process_group(a.dist_backend)

    pkg = entry.load_package()
    scene = pkg.scene_by_name(a.scene)
    cam = pkg.cornell_camera(a.width, a.height)
    backward = not a.forward_only
    r = pkg.HipRenderer(local_rank)          # raises without libdrt_hip.so / a device: no fallback
    # a render loop: the scene's own path kernel is compiled at the first frame (setup, before the warm-up; the library would
    # do it by itself after 2^31 path-bounces -- DRT_SPECIALISE_AUTO --; `program` / `specialise_ms` in the line say what ran)
    r.set_specialisation(pkg.SPECIALISE_NOW)
    r.upload_scene(scene)
    dev = torch.device("cuda", local_rank)

    # the ONE collective of the path lives behind the C ABI: rank 0 makes the id, the launcher's rendezvous
    # (torch.distributed here) hands it to the other ranks, every rank joins with its own context
    reduce_mode = None
    if use_dist and backward:
        reduce_mode = a.reduce
        if reduce_mode == "library" and not load_distributed().join_library_communicator(r, pkg):
            # The ONE collective of the path belongs to the library: a run that cannot make its communicator over the ranks it
            # was given must not quietly measure something else.  (--same-gpu, the single-GPU plumbing test, is the one place
            # where RCCL is known to refuse -- two ranks on one device -- and torch.distributed stands in.)
            if not a.same_gpu:
                print(f"bench.py: rank {rank}: the in-library RCCL communicator over {world} ranks could not be made "
                      f"(drt_hip_comm_init_rank); refusing to fall back to torch.distributed -- pass --reduce torch to measure that",
                      file=sys.stderr)
                sys.exit(3)
            if rank == 0:
                print("bench.py: --same-gpu: in-library communicator unavailable; reducing with torch.distributed", file=sys.stderr)
            reduce_mode = "torch"
    if reduce_mode == "library" and r.comm_size != world:
        print(f"bench.py: rank {rank}: the library's communicator has {r.comm_size} ranks, the job {world}", file=sys.stderr)
        sys.exit(3)
    # the library's all-reduce runs on the context's second stream and overlaps the next step's kernels
    # (DRT_RENDER_ALLREDUCE_ASYNC; the steps alternate between two gradient buffers, the fence waits for both streams)
    flags = (pkg.RENDER_UNBIASED if (a.unbiased and backward) else 0) | \
            ((pkg.RENDER_ALLREDUCE_ASYNC if a.allreduce == "async" else pkg.RENDER_ALLREDUCE) if reduce_mode == "library" else 0)
    rp = pkg.RenderParams(spp=a.spp * world, min_bounces=a.min_bounces, absorb=a.absorb, seed=1,
                          shard=rank, n_shards=world, band_rows=16, batch_paths=a.batch_paths,
                          flags=flags, bounces_per_launch=a.bounces_per_launch)

    out_rgb = torch.zeros((a.height, a.width, 3), dtype=torch.float32, device=dev)
    grads = [torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in range(2)]
    # reduce_mode == "torch" (testing / A-B): the all-reduce of step i runs on its own stream while step i + 1
    # renders into the other gradient buffer
    ext = torch.cuda.ExternalStream(r.stream, device=dev) if reduce_mode == "torch" else None
    comm = torch.cuda.Stream(device=dev) if reduce_mode == "torch" else None
    reduced = [None, None]
    n_step = [0]

    def step(timing=False, params=None):
        b = n_step[0] & 1
        n_step[0] += 1
        grad = grads[b]
        if reduced[b] is not None:
            ext.wait_event(reduced[b])         # step i - 2's all-reduce read this buffer
        st = r.render_device(cam, params or rp, out_rgb.data_ptr(), grad.data_ptr() if backward else 0,
                             backward=backward, timing=timing, sync=False)
        if reduce_mode == "torch":
            rendered = torch.cuda.Event()
            rendered.record(ext)
            comm.wait_event(rendered)
            with torch.cuda.stream(comm):
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)
                done = torch.cuda.Event()
                done.record(comm)
            reduced[b] = done
        return st

    def fence():
        """device work of this rank done, then every rank here -> the time at which THIS rank's work was done"""
        r.synchronize()
        torch.cuda.synchronize(dev)
        t_done = time.perf_counter()
        if use_dist:
            dist.barrier()
        return t_done

    # Setup, before the W warm-up steps: the device comes out of idle at low clocks and takes some tens of milliseconds of
    # load to reach its operating point (measured on config 3, ms per frame over 5 / 10 / 20 frames from a cold start: 0.865 /
    # 0.803 / 0.758, then flat) -- W = 5 steps of 0.8 ms do not get it there, and the timed region of K = 20 steps would
    # measure the ramp.  So the loop is driven for --preheat-ms (default 60) first; reported in the line (`preheat_ms`).
    preheat_frames = 0
    if a.preheat_ms > 0:
        # (the same number of frames on every rank -- each carries a collective: one frame is timed, the slowest rank's time
        #  decides)
        step()
        fence()
        tp = time.perf_counter()
        step()
        fence()
        t_frame = time.perf_counter() - tp
        if use_dist:
            tt = torch.tensor([t_frame], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_frame = float(tt.item())
        preheat_frames = int(max(1, min(400, a.preheat_ms * 1e-3 / max(t_frame, 1e-6))))
        for i in range(preheat_frames):
            step()
            if (i + 1) % 8 == 0:
                fence()
        fence()
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    # The K steps stand between a barrier + device synchronisation on either side; a rank's time runs from the opening
    # barrier to the moment ITS device work is done, and the MAX over the ranks is the job's time -- the closing barrier's own
    # latency (a host round trip through the process group: ~1 ms here, 8 % of twenty 0.66-ms steps) is not rendering time.
    elapsed = fence() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel HIP-event timing (events on the context's own stream), same workload, same
    # process, right after the timed region; also yields the segment count of one step
    kernel_ms = {k: 0.0 for k in pkg.KERNEL_NAMES}
    kernel_launches = {k: 0 for k in pkg.KERNEL_NAMES}
    # (a timed step ends in a host round trip -- the event times are read back -- and the device idles meanwhile: an untimed
    #  step in front of every timed one, enqueued without waiting, keeps the timed kernels in the state the timed region's
    #  kernels ran in; measured without it on one box: 0.736 ms against the trace's 0.696 for the same kernel)
    n_prof = max(1, min(a.steps, 10))
    stats = None
    for _ in range(n_prof):
        step()
        stats = step(timing=True)
        for k in pkg.KERNEL_NAMES:
            kernel_ms[k] += stats["kernels"][k]["ms"]
            kernel_launches[k] += stats["kernels"][k]["launches"]
    fence()
    segments = stats["segments"]
    paths = stats["paths"]
    queue_rays = stats["queue_rays_read"] + stats["queue_rays_written"]
    devices = None
    if use_dist:
        t = torch.tensor([segments, paths], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_segments, total_paths = float(t[0].item()), float(t[1].item())
        props = torch.cuda.get_device_properties(local_rank)
        # (what the LIBRARY's context renders on and how many ranks ITS communicator spans -- not what torch saw)
        mine = f"rank {rank}: cuda:{local_rank} pci {r.pci_bus_id()} (torch: {getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'pci_device_id', '?')}), library communicator of {r.comm_size} ranks"
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        # First contact with a multi-GPU node must not be mis-measured: N ranks = N DIFFERENT devices as the library sees them
        # (PCI bus ids of the contexts' devices), unless --same-gpu asked for the single-GPU plumbing run.
        pcis = [None] * world
        dist.all_gather_object(pcis, r.pci_bus_id())
        if len(set(pcis)) != world and not a.same_gpu:
            if rank == 0:
                print(f"bench.py: {world} ranks render on {len(set(pcis))} distinct devices ({pcis}); every rank needs a GPU of its own "
                      f"(pass --same-gpu for the single-GPU plumbing run)", file=sys.stderr)
            sys.exit(3)
        if rank == 0:
            for line_ in devices:
                print("bench.py:", line_, file=sys.stderr)
    else:
        total_segments, total_paths = float(segments), float(paths)

    ms_per_step = elapsed / a.steps * 1e3
    value = total_segments * a.steps / elapsed * 1e-6   # Mray/s, whole job

    # ---- per-kernel algorithmic bytes and the roofline of the dominant kernel ----------
    # units: what the library counted for each kernel (drt_hip_stats.units): paths (K1, K5), segments (K3, K6, k_path),
    # the rays k_intersect tested (mesh scenes: camera rays only), the candidate rays the BVH walk took
    units = {k: stats["kernels"][k]["units"] for k in pkg.KERNEL_NAMES}
    # fused shade: tape 8 B/segment + 32 B per ray read from / written to the queue (the library counts both);
    # unfused (mesh scenes): ray 24 + id 8 + hit 8 read, ray 24 + id 8 + tape 8 written;
    # k_path: a path lives in registers from the eye to its end; the launch writes 24 B per pixel and sample range,
    # 192 B per block (gradient partials) and 8 B per wave (counters): stats["path_bytes"], ~0.4 B per segment;
    # k_intersect_mesh: per CANDIDATE ray a 36-byte record (slot, origin, direction, analytic t, tie-break index) read and
    # at most one 8-byte hit written (the BVH itself is L2 / Infinity-Cache resident)
    seg = max(1, segments)
    # mesh scenes, k_shade<TAIL> (no k_intersect launch: the shade launch intersects the ray it produces with the analytic shapes
    # itself): the unfused 80 B + the next depth's hit lane (8 B written) + a 40-byte candidate record per ray that reaches the
    # mesh bounds (the BVH walk's input: units["intersect_mesh"] of them per step)
    shade_tail_bytes = 88.0 + 40.0 * units["intersect_mesh"] / seg
    bpu = {"intersect": 32.0, "intersect_mesh": 44.0,
           "shade": 80.0 if kernel_launches["intersect"] else
                    (shade_tail_bytes if kernel_launches["intersect_mesh"] else
                     (8.0 if kernel_launches["backward"] else 16.0 * paths / seg) + 32.0 * queue_rays / seg),
           "path": stats.get("path_bytes", 0) / seg,
           "backward": 8.0, "raygen": 32.0, "gradreduce": 0.0,
           # K5 reads 16 B of radiance per path; behind k_path it reads the per-range pixel sums that launch wrote
           "film": stats.get("path_bytes", 0) / max(1, paths) if kernel_launches["path"] else 16.0}
    per_kernel = {}
    for k in pkg.KERNEL_NAMES:
        ms = kernel_ms[k] / n_prof
        if kernel_launches[k] == 0:
            continue
        gbs = units[k] * bpu[k] / (ms * 1e-3) * 1e-9 if ms > 0 else 0.0
        per_kernel[k] = {"ms_per_step": round(ms, 4), "launches_per_step": kernel_launches[k] // n_prof,
                         "units_per_step": int(units[k]), "bytes_per_unit": round(bpu[k], 3), "achieved_GBs": round(gbs, 1),
                         "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    dominant = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"]) if per_kernel else "shade"
    # rocprofv3 PMC figures of the dominant kernel (profiles/traffic.json, written by tools/profile.sh, one entry per
    # profiled workload): quoted only for the very workload this run measures
    pmc = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    depth_text = f"roulette from bounce {a.min_bounces} with absorption {a.absorb:g}" if a.roulette else f"depth {a.depth}"
    depth_key = f"rr{a.absorb:g}b{a.min_bounces}" if a.roulette else f"d{a.depth}"
    mode_key = "unbiased" if a.unbiased else ("fwd" if a.forward_only else "fwdbwd")
    workload_key = f"{a.scene}:{a.width}x{a.height}x{a.spp}:{depth_key}:{mode_key}"
    pmc_note = None
    if os.path.exists(tpath) and a.bounces_per_launch == 0:
        try:
            tj = json.load(open(tpath))
            wl = tj.get("workloads") or {}
            entry_w = wl.get(workload_key) or (tj if tj.get("workload") == workload_key else {})
            pmc = dict(entry_w.get(dominant) or {})
            if not pmc:
                # not profiled at this size: the same scene, depth and mode at ANOTHER frame size -- instructions and bytes per
                # UNIT of the kernel's work (segments, candidate rays) do not depend on the frame
                my_units = per_kernel.get(dominant, {}).get("units_per_step", 0) / max(1, per_kernel.get(dominant, {}).get("launches_per_step", 1))
                for key, ent in sorted(wl.items()):
                    sc, dims, dk_, md = key.split(":")
                    e2 = ent.get(dominant) or {}
                    if sc == a.scene and dk_ == depth_key and md == mode_key and e2.get("units_per_launch") and my_units > 0:
                        scale = my_units / float(e2["units_per_launch"])
                        pmc = {k2: (v2 * scale if k2.endswith("_per_launch") or k2 in ("fetch_raw_bytes", "write_bytes") else v2)
                               for k2, v2 in e2.items() if isinstance(v2, (int, float))}
                        pmc_note = f"PMC counts of '{key}' carried over per unit of work ({scale:g} x per launch): this size was not profiled"
                        break
        except Exception:
            pmc = {}
    dk = per_kernel.get(dominant, {"achieved_GBs": 0.0, "launches_per_step": 1, "ms_per_step": 0.0})
    launch_ms = dk["ms_per_step"] / max(1, dk["launches_per_step"])
    register_resident = dominant == "path" or (dominant == "shade" and not kernel_launches["intersect"]
                                               and dk["launches_per_step"] < a.depth * max(1, stats["batches"]) and not a.unbiased)
    traffic = round(pmc["bytes_per_launch"] / (launch_ms * 1e-3) * 1e-9, 1) if pmc.get("bytes_per_launch") and launch_ms > 0 else None
    pmc_launch_us = pmc.get("avg_launch_us")
    hbm_view = {"achieved": dk["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["achieved_GBs"] / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_roof": round(dk["achieved_GBs"] / HBM_COPY_GBS, 4),
                "note": "algorithmic bytes of the dominant kernel (units x bytes per unit) over its HIP-event time"}
    valu_view = None
    # The instruction count is a STORED figure (profiles/traffic.json): it describes this run only if the kernel is still the
    # one that was profiled.  The profile also recorded the kernel's launch time: where the live time is off by more than 5 %
    # (boxes of the pool differ by ~3 %) the count is withheld instead of being divided by a time it does not belong to.
    # Better than a time: the profile records WHICH kernels it counted (a hash of the device sources the library was built from) and the
    # work of a launch (segments: a pure function of workload and seed).  Same sources and same work = the same instructions, whatever the
    # box's clock does today; the time check remains for profiles that carry no hash.
    stale = None
    same_kernels = None
    if pmc.get("kernel_sources_sha16") and not pmc_note:
        live_units = dk.get("units_per_step", 0) / max(1, dk["launches_per_step"]) if dk.get("units_per_step") else None
        same_kernels = (pmc["kernel_sources_sha16"] == entry.kernel_sources_sha16() and bool(live_units) and bool(pmc.get("units_per_launch")))
        if same_kernels and float(pmc["units_per_launch"]) != float(live_units):
            # the same kernels on the same scene, depth and mode, but another share of the frame (a rank of an N-GPU run renders other
            # rows at N x the samples: a few segments more or less): instructions and bytes per SEGMENT are what carries over
            scale = float(live_units) / float(pmc["units_per_launch"])
            if 0.8 <= scale <= 1.25:
                pmc = {k2: (v2 * scale if isinstance(v2, (int, float)) and (k2.endswith("_per_launch") or k2 in ("fetch_raw_bytes", "write_bytes")) else v2)
                       for k2, v2 in pmc.items()}
                traffic = round(pmc["bytes_per_launch"] / (launch_ms * 1e-3) * 1e-9, 1) if pmc.get("bytes_per_launch") and launch_ms > 0 else None
            else:
                same_kernels = False
    if pmc.get("valu_insts_per_launch") and pmc.get("avg_launch_us") and launch_ms > 0 and not pmc_note and not same_kernels:
        off = launch_ms * 1e3 / float(pmc["avg_launch_us"]) - 1.0
        if abs(off) > 0.05:
            stale = (f"profiles/traffic.json holds k_{dominant} at {float(pmc['avg_launch_us']):.1f} us per launch, this run measures "
                     f"{launch_ms * 1e3:.1f} us ({off:+.1%}): the kernel (or the box) is not the profiled one -- instruction counts "
                     f"withheld; tools/profile.sh regenerates the profile")
            pmc_note = stale
    if pmc.get("valu_insts_per_launch") and launch_ms > 0 and not stale:
        ginst = pmc["valu_insts_per_launch"] / (launch_ms * 1e-3) * 1e-9
        valu_view = {"achieved": round(ginst, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
                     "frac": round(ginst / VALU_PEAK_GINST, 4),
                     "frac_of_measured_issue_roof": round(ginst / VALU_MEASURED_GINST, 4),
                     "valu_insts_per_launch": pmc.get("valu_insts_per_launch"),
                     "salu_insts_per_launch": pmc.get("salu_insts_per_launch"),
                     "count_valid_because": ("the profile was taken of these very kernels (hash of the device sources "
                                             f"{pmc.get('kernel_sources_sha16')}) on the same work per launch" if same_kernels else
                                             "the profile's own launch time is within 5 % of the live one"),
                     "note": "SQ_INSTS_VALU per launch (profiles/traffic.json, rocprofv3 --pmc on this workload) / live launch "
                             "time; peak = one wave64 VALU op per 2 cycles per SIMD-32, 256 CUs x 4 SIMDs x 2.4 GHz"}
    # The top-level achieved / peak / unit / frac describe the roof that BOUNDS the dominant kernel: HBM bytes for the
    # streaming kernels; vector-instruction issue for the register-resident k_path (it moves ~0.4 B per segment) and
    # for the BVH walk (divergent, latency-bound: neither roof is near -- both fractions are printed).  The other
    # view stays beside it (`hbm` / `valu`).
    issue_bound = register_resident or dominant == "intersect_mesh"
    # (no rocprofv3 PMC profile of this workload in profiles/traffic.json -- tools/profile.sh makes one --: the HBM view, which
    #  needs none, is what the top level carries then)
    top = valu_view if (issue_bound and valu_view) else hbm_view
    roofline = {"bound": ("valu-issue (paths are register-resident: the kernel moves almost no bytes)" if register_resident
                          else ("the CU's L1 path (divergent BVH walk out of L2; by the TCP's and TD's own counters, profiles/r06_walk_counters.txt: "
                                "the L1 clocked 93 % of a launch, 31 % of the cycles stalled behind lines in flight, 780 cycles per wave-load against "
                                "230 for an L2 hit, the data-return unit 86 % busy) -- neither roof is near: vector issue 0.37 of nominal (SQ_INSTS_VALU "
                                "x 2 cycles, 37 of 64 lanes per instruction; the fraction printed here), HBM 0.10" if dominant == "intersect_mesh" else "hbm")),
                "kernel": "k_" + dominant, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
                "frac": top["frac"],
                "traffic": traffic,
                "traffic_note": "GB/s of PMC-counted HBM bytes per launch (profiles/traffic.json, rocprofv3 --pmc, FETCH_SIZE "
                                "doubled per the gfx950 correction) over the live launch time" if traffic else None,
                "avg_launch_ms": round(launch_ms, 4),
                "mode": "one launch at a time: HIP events around every launch of frames rendered in stream order (as rocprofv3 "
                        "sees them with DRT_HIP_OVERLAP_FRAMES=0, profiles/); `value` pipelines two frames, see `pipelined`",
                "pmc_note": pmc_note, "pmc_launch_us": pmc_launch_us,
                "hbm": hbm_view, "valu": valu_view,
                "kernels": per_kernel}
    if dominant == "path" and valu_view and dk["launches_per_step"] > 0:
        # the chip-level issue rate of the HEADLINE mode: the same instructions per step over the step time of the timed region
        # (the grids of two consecutive frames overlap: a launch's own begin-to-end time is longer than a step there)
        g2 = pmc["valu_insts_per_launch"] * dk["launches_per_step"] / (ms_per_step * 1e-3) * 1e-9
        roofline["pipelined"] = {"achieved": round(g2, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
                                 "frac": round(g2 / VALU_PEAK_GINST, 4),
                                 "frac_of_measured_issue_roof": round(g2 / VALU_MEASURED_GINST, 4), "ms_per_step": round(ms_per_step, 4)}
    if dominant == "intersect_mesh" and pmc.get("lane_stats"):
        roofline["walk_lanes"] = pmc["lane_stats"]      # tools/bvh_stats.py on the same scene (a -DDRT_BVH_STATS build)

    extra = rank == 0 and world == 1 and not a.no_extra_views
    import dataclasses
    if extra and register_resident and not a.unbiased:
        # The HBM roofline describes the STREAMING wavefront: the same workload with one shade launch per bounce
        # (rays through the queue in HBM after every bounce), timed live.
        rp1 = dataclasses.replace(rp, bounces_per_launch=1)
        for _ in range(2):
            step(params=rp1)
        fence()
        t4 = time.perf_counter()
        for _ in range(n_prof):
            step(params=rp1)
        fence()
        dt4 = (time.perf_counter() - t4) / n_prof
        ms1, q1 = 0.0, 0
        for _ in range(n_prof):
            st1 = step(timing=True, params=rp1)
            ms1 += st1["kernels"]["shade"]["ms"]
            q1 = st1["queue_rays_read"] + st1["queue_rays_written"]
        fence()
        ms1 /= n_prof
        bpu1 = 8.0 + 32.0 * q1 / seg
        gbs1 = segments * bpu1 / (ms1 * 1e-3) * 1e-9 if ms1 > 0 else 0.0
        roofline["streaming"] = {"bounces_per_launch": 1, "value": round(segments / dt4 * 1e-6, 2), "unit": "Mray/s",
                                 "ms_per_step": round(dt4 * 1e3, 4), "kernel": "k_shade", "shade_ms_per_step": round(ms1, 4),
                                 "bytes_per_unit": round(bpu1, 3), "achieved": round(gbs1, 1), "peak": HBM_PEAK_GBS,
                                 "frac": round(gbs1 / HBM_PEAK_GBS, 4),
                                 "frac_of_measured_copy_roof": round(gbs1 / HBM_COPY_GBS, 4)}

    def timed_variant(**kw):
        """Mray/s of a variant of the headline call (device buffers, same frame), a few steps."""
        bw = kw.pop("backward", backward)
        f64 = kw.pop("f64", False)
        unb = kw.pop("unbiased", False)
        rpv = dataclasses.replace(rp, flags=(rp.flags | (pkg.RENDER_F64 if f64 else 0) | (pkg.RENDER_UNBIASED if unb else 0)))

        def one(timing=False):
            return r.render_device(cam, rpv, out_rgb.data_ptr(), grads[0].data_ptr() if bw else 0, backward=bw,
                                   timing=timing, sync=False)
        one()
        fence()
        n = 2 if f64 else n_prof
        t5 = time.perf_counter()
        for _ in range(n):
            one()
        fence()
        dt5 = (time.perf_counter() - t5) / n
        segs = one(timing=True)["segments"]
        fence()
        return {"value": round(segs / dt5 * 1e-6, 2), "unit": "Mray/s", "ms_per_step": round(dt5 * 1e3, 4)}

    # ---- what the headline is made of.  `value` = frames that do not depend on each other, the path kernels of two consecutive
    # ones overlapping (device-pointer renders without a wait).  `serial_frame`: the same frames with DRT_RENDER_SERIAL --
    # everything of a frame on the context's stream, in order: what an optimisation loop gets, whose frame i + 1 needs frame
    # i's gradients.  `generic_program`: serial frames on a context that reads the shape kinds at run time
    # (DRT_SPECIALISE_GENERIC) instead of running the kernel compiled for the scene: what specialisation buys.
    serial_view = generic_view = None
    if rank == 0 and world == 1 and not a.no_extra_views:      # (profiling runs keep their kernel statistics unmixed)
        def frames(rr, params, n):
            for _ in range(3):
                rr.render_device(cam, params, out_rgb.data_ptr(), grads[0].data_ptr() if backward else 0, backward=backward, sync=False)
            rr.synchronize()
            torch.cuda.synchronize(dev)
            t7 = time.perf_counter()
            for _ in range(n):
                rr.render_device(cam, params, out_rgb.data_ptr(), grads[0].data_ptr() if backward else 0, backward=backward, sync=False)
            rr.synchronize()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t7) / n
        rps = dataclasses.replace(rp, flags=rp.flags | pkg.RENDER_SERIAL)
        dts = frames(r, rps, a.steps)
        serial_view = {"value": round(total_segments / dts * 1e-6, 2), "unit": "Mray/s", "ms_per_step": round(dts * 1e3, 4),
                       "note": "DRT_RENDER_SERIAL: frames in stream order, no overlap between consecutive frames' path kernels"}
        if stats.get("path_program") in ("builtin", "specialised"):
            try:
                rg = pkg.HipRenderer(local_rank)
                rg.set_specialisation(pkg.SPECIALISE_GENERIC)
                rg.upload_scene(scene)
                dtg = frames(rg, rps, a.steps)
                generic_view = {"value": round(total_segments / dtg * 1e-6, 2), "unit": "Mray/s", "ms_per_step": round(dtg * 1e3, 4),
                                "note": "serial frames with the shape kinds read at run time (the kind-sorted program in LDS; "
                                        "DRT_SPECIALISE_GENERIC) -- bit-identical results"}
                rg.close()
            except Exception as exc:
                generic_view = {"error": f"{type(exc).__name__}: {exc}"}

    # `many_parameters`: config 3's frame with an albedo per SHAPE of the reference's scene (10 parameters instead of 4; the reference
    # differentiates with respect to any number, vector.hpp:185-191): serial frames through the one-launch kernel's general form
    # (vertex history + per-wave gradient tables) -- until round 6 such a scene took the queue wavefront (2.40 ms per frame)
    many_view = None
    if rank == 0 and world == 1 and not a.no_extra_views and a.is_config and a.config == 3 and backward:
        try:
            sm = pkg.scene_by_name("cornell_shapes")
            rm = pkg.HipRenderer(local_rank)
            rm.set_specialisation(pkg.SPECIALISE_NOW)
            rm.upload_scene(sm)
            gm = torch.zeros((sm.n_params, 3), dtype=torch.float64, device=dev)
            rps_m = dataclasses.replace(rp, flags=rp.flags | pkg.RENDER_SERIAL)
            for _ in range(5):
                rm.render_device(cam, rps_m, out_rgb.data_ptr(), gm.data_ptr(), backward=True, sync=False)
            rm.synchronize(); torch.cuda.synchronize(dev)
            tm = time.perf_counter()
            for _ in range(a.steps):
                rm.render_device(cam, rps_m, out_rgb.data_ptr(), gm.data_ptr(), backward=True, sync=False)
            rm.synchronize(); torch.cuda.synchronize(dev)
            dtm = (time.perf_counter() - tm) / a.steps
            stm = rm.render_device(cam, rps_m, out_rgb.data_ptr(), gm.data_ptr(), backward=True, timing=True)
            many_view = {"scene": "cornell_shapes", "n_params": sm.n_params, "value": round(stm["segments"] / dtm * 1e-6, 2), "unit": "Mray/s",
                         "ms_per_step": round(dtm * 1e3, 4), "launches": {k: v["launches"] for k, v in stm["kernels"].items() if v["launches"]},
                         "note": "serial frames (DRT_RENDER_SERIAL), an albedo parameter per shape of render.cpp's scene; compare `serial_frame` "
                                 "(4 parameters); profiles/r06_param_cliff_before.txt holds the 2.40 ms of the queue wavefront"}
            rm.close()
        except Exception as exc:
            many_view = {"error": f"{type(exc).__name__}: {exc}"}

    f64_view = fwd_view = unb_view = None
    if extra and not a.unbiased and backward and world == 1:
        unb_view = dict(timed_variant(unbiased=True), note="the same frame with the reference's UNBIASED integration operator "
                                                           "(integrate.hpp:39-52: a fresh suffix path per vertex, O(depth^2) segments)")
    if extra and not a.unbiased:
        f64_view = dict(timed_variant(f64=True), roofline={
            "bound": "f64 vector issue", "kernel": "k_path<double>", "valu_per_launch": 9.208e8, "launch_ms_profiled": 1.707,
            "achieved": 539.0, "unit": "G wave-instr/s", "peak_all_f64": 614.4, "peak_all_f32": 1228.8, "frac_of_f64_rate": 0.88,
            "frac_of_f32_rate": 0.44, "waves_per_simd": 4,
            "source": "profiles/r06_f64_counters.txt (rocprofv3 --pmc SQ_INSTS_VALU over tools/f64_frames.py: a STORED count, "
                      "valid for config 3's frame only)"} if a.is_config and a.config == 3 else None, note="the same call with DRT_RENDER_F64: every kernel computes and stores in "
                                                      "double, the reference's precision (render.cpp:22)")
        if backward:
            fwd_view = dict(timed_variant(backward=False), note="forward only (BASELINE config 2 when the headline is config 3)")

    # Two contexts on this device rendering the headline's frames alternately (no collective): the last, partly filled round of
    # one frame's k_path grid is filled by the next frame's first -- what a render loop over several views gets (HISTORY.md 1)
    two_ctx_view = None
    if extra and register_resident and not a.unbiased and world == 1:
        try:
            r2 = pkg.HipRenderer(0)
            r2.upload_scene(scene)
            prp = pkg.RenderParams(spp=a.spp, min_bounces=a.min_bounces, absorb=a.absorb, seed=1, batch_paths=a.batch_paths)
            out2 = torch.zeros_like(out_rgb)
            g2 = torch.zeros_like(grads[0])
            pair = ((r, out_rgb, grads[0]), (r2, out2, g2))

            def both(n):
                for i in range(n):
                    rr, oo, gg = pair[i & 1]
                    rr.render_device(cam, prp, oo.data_ptr(), gg.data_ptr() if backward else 0, backward=backward)
                r.synchronize()
                r2.synchronize()
            both(4)
            n2 = max(4, min(2 * a.steps, 40))
            t6 = time.perf_counter()
            both(n2)
            dt6 = (time.perf_counter() - t6) / n2
            two_ctx_view = {"value": round(total_segments / dt6 * 1e-6, 2), "unit": "Mray/s", "ms_per_step": round(dt6 * 1e3, 4),
                            "note": "the same frames rendered alternately by TWO contexts on this device (each on its own stream, "
                                    "no collective): consecutive k_path grids overlap"}
            r2.close()
        except Exception as exc:
            two_ctx_view = {"error": f"{type(exc).__name__}: {exc}"}

    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        oracle = entry.load_oracle()
        # a bounded sample of the same workload: a pilot measures the oracle's rate on a small frame of the same
        # scene and depth, then the sample is sized for ~cpu_seconds (mesh scenes are brute force on the CPU:
        # their sample is a handful of pixels)
        pw, ph, ps = 32, 32, 1
        t1 = time.perf_counter()
        pilot = oracle.render(scene, pkg.cornell_camera(pw, ph), pkg.RenderParams(spp=ps, min_bounces=a.min_bounces, absorb=a.absorb, seed=1),
                              backward=backward)
        rate = pilot["stats"]["segments"] / max(1e-6, time.perf_counter() - t1)      # rays / s
        per_path = pilot["stats"]["segments"] / (pw * ph * ps)
        budget_paths = max(1.0, a.cpu_seconds * rate / per_path)
        cw, ch, cs = a.width, a.height, a.spp
        if budget_paths >= cw * ch:
            cs = int(max(1, min(a.spp, budget_paths // (cw * ch))))
        else:                               # not even one sample per pixel: shrink the frame (same aspect)
            f = (budget_paths / (cw * ch)) ** 0.5
            cw, ch, cs = max(8, int(cw * f)), max(8, int(ch * f)), 1
        ccam = pkg.cornell_camera(cw, ch)
        crp = pkg.RenderParams(spp=cs, min_bounces=a.min_bounces, absorb=a.absorb, seed=1)
        t1 = time.perf_counter()
        ref = oracle.render(scene, ccam, crp, backward=backward)
        dt = time.perf_counter() - t1
        cpu_baseline = {"value": float(f"{ref['stats']['segments'] / dt * 1e-6:.4g}"), "unit": "Mray/s",
                        "cores": 1, "kind": "port",
                        "sample": f"same scene, frame {cw}x{ch} of {a.width}x{a.height}, {cs} of the {a.spp} spp, "
                                  f"{depth_text}, {'fwd+bwd' if backward else 'fwd'}: "
                                  f"{ref['stats']['segments']} rays in {dt:.1f} s (fp64 C restatement, 1 thread)"}
        # parity of this very workload's gradients against the CPU restatement (same RNG keys): run the
        # device once more on the sample's frame and spp
        if backward:
            img_d, g_d, _ = r.render(ccam, crp, backward=True)
            gerr = float(np.abs(g_d - ref["grads"]).max() / np.abs(ref["grads"]).max())
            cpu_baseline["grad_max_rel_err_vs_cpu"] = gerr
        # ... and THE REFERENCE ITSELF where its binary travelled with the snapshot (oracle/_ref/ref_harness: the unmodified
        # reference headers behind oracle/ref_harness.cpp's driver, compiled in the build container by oracle/Makefile):
        # the same workload on a sample sized for ~ref_seconds, timed by the harness around its render loop, and the
        # device's gradients of that very sample against the reference's own backward()
        if oracle.have_reference() and a.ref_seconds > 0:
            try:
                pw2 = 8 if scene.meshes else 32
                prp = pkg.RenderParams(spp=1, min_bounces=a.min_bounces, absorb=a.absorb, seed=1)
                pil = oracle.render_reference(scene, pkg.cornell_camera(pw2, pw2), prp, backward=backward)
                rate2 = pil["stats"]["segments"] / max(1e-6, pil["stats"]["seconds"])
                per_path2 = max(1e-9, pil["stats"]["segments"] / (pw2 * pw2))
                budget2 = max(1.0, a.ref_seconds * rate2 / per_path2)
                rw, rh, rs_ = a.width, a.height, 1
                if budget2 >= rw * rh:
                    rs_ = int(max(1, min(a.spp, budget2 // (rw * rh))))
                else:
                    f2 = (budget2 / (rw * rh)) ** 0.5
                    rw, rh = max(8, int(rw * f2)), max(8, int(rh * f2))
                rcam = pkg.cornell_camera(rw, rh)
                rrp = pkg.RenderParams(spp=rs_, min_bounces=a.min_bounces, absorb=a.absorb, seed=1)
                res = oracle.render_reference(scene, rcam, rrp, backward=backward)
                rsec = max(1e-6, res["stats"]["seconds"])
                cpu_baseline["reference"] = {
                    "value": float(f"{res['stats']['segments'] / rsec * 1e-6:.4g}"), "unit": "Mray/s", "cores": 1, "kind": "reference",
                    "sample": f"same scene, frame {rw}x{rh} of {a.width}x{a.height}, {rs_} of the {a.spp} spp, {depth_text}, "
                              f"{'fwd+bwd' if backward else 'fwd'}: {res['stats']['segments']} rays in {rsec:.1f} s "
                              f"(the unmodified reference headers, double, 1 thread; oracle/_ref/ref_harness)"}
                img_r, g_r, _ = r.render(rcam, rrp, backward=backward)
                cpu_baseline["reference"]["image_max_abs_err_vs_reference"] = float(np.abs(img_r - res["image"]).max())
                if backward:
                    cpu_baseline["reference"]["grad_max_rel_err_vs_reference"] = float(
                        np.abs(g_r - res["grads"]).max() / max(1e-300, np.abs(res["grads"]).max()))
            except Exception as exc:        # (the checker's binary is optional; the port above is the baseline)
                cpu_baseline["reference"] = {"error": f"{type(exc).__name__}: {exc}"}
                print(f"bench.py: cpu_baseline.reference failed: {type(exc).__name__}: {exc}", file=sys.stderr)
        elif a.ref_seconds > 0:
            print("bench.py: oracle/_ref/ref_harness is not there (a clean checkout: it is built in the build container from "
                  "/root/reference by oracle/Makefile and travels as a binary) -- no cpu_baseline.reference", file=sys.stderr)
        if a.cpu_all_cores:
            # the reference is single-threaded by construction (global rand()); this is N independent
            # processes, each rendering its interleaved row bands of the same sample
            import multiprocessing as mp
            n = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
            jobs = [(a.scene, cw, ch, cs, a.min_bounces, a.absorb, backward, i, n) for i in range(n)]
            with mp.get_context("spawn").Pool(n) as pool:
                # start the workers and load the checker in each of them before the clock starts
                pool.map(_oracle_shard, [(a.scene, 16, 16, 1, 1, False, 0, 1)] * n, chunksize=1)
                t2 = time.perf_counter()
                segs = sum(pool.map(_oracle_shard, jobs, chunksize=1))
                dt2 = time.perf_counter() - t2
            cpu_baseline["all_cores"] = {"value": round(segs / dt2 * 1e-6, 2), "unit": "Mray/s", "cores": n,
                                         "note": "N independent row-band processes of the same sample, workers warmed up; N = usable CPUs reported by the OS (a container CPU quota may be lower)"}

    # the same frames through the HOST-BUFFER entry points (image and gradients copied to pageable host memory over
    # PCIe): reported beside `value`, never as `value`.  `host_buffers`: drt_hip_render_async / drt_hip_wait, two frames
    # in flight (frame i's copy overlaps frame i + 1's kernels: what a render loop uses); `sync`: drt_hip_render, which
    # returns with the results on the host (a copy and a stream synchronisation per frame, the GPU idle meanwhile)
    host_buffers = None
    if extra:
        hrp = pkg.RenderParams(spp=a.spp, min_bounces=a.min_bounces, absorb=a.absorb, seed=1, batch_paths=a.batch_paths)
        n_host = max(2, min(2 * a.steps, 40))     # (a pipeline: enough frames that its fill and drain do not set the rate)
        unb = a.unbiased and backward
        # (the synchronous call the way a render loop makes it: the same image buffer every frame, no statistics --
        #  in pageable memory, and pinned once with drt_hip_pin_host so that the finishing kernel stores the image straight into it)
        sync_img = np.zeros((a.height, a.width, 3), dtype=np.float32)
        r.render(cam, hrp, backward=backward, unbiased=unb, img_out=sync_img, want_stats=False)
        t3 = time.perf_counter()
        for _ in range(n_host):
            r.render(cam, hrp, backward=backward, unbiased=unb, img_out=sync_img, want_stats=False)
        dt3 = (time.perf_counter() - t3) / n_host
        r.pin_host(sync_img)
        r.render(cam, hrp, backward=backward, unbiased=unb, img_out=sync_img, want_stats=False)
        t3 = time.perf_counter()
        for _ in range(n_host):
            r.render(cam, hrp, backward=backward, unbiased=unb, img_out=sync_img, want_stats=False)
        dt3p = (time.perf_counter() - t3) / n_host
        r.unpin_host(sync_img)
        r.wait(r.render_async(cam, hrp, backward=backward, unbiased=unb))
        n_fly = pkg.FRAMES_IN_FLIGHT
        bufs = [(np.zeros((a.height, a.width, 3), dtype=np.float32), np.zeros((scene.n_params, 3), dtype=np.float64)) for _ in range(n_fly)]
        for im, _g in bufs:
            im.fill(0.0)                  # (touched once: a render loop keeps a set of buffers per frame in flight)
        t3 = time.perf_counter()
        flying = []
        for it in range(n_host):
            flying.append(r.render_async(cam, hrp, backward=backward, unbiased=unb, img_out=bufs[it % n_fly][0],
                                         grads_out=bufs[it % n_fly][1] if backward else None))
            if len(flying) == n_fly:
                r.wait(flying.pop(0), want_stats=False)
        while flying:
            r.wait(flying.pop(0), want_stats=False)
        dt3a = (time.perf_counter() - t3) / n_host
        host_buffers = {"value": round(total_segments / dt3a * 1e-6, 2), "unit": "Mray/s",
                        "ms_per_step": round(dt3a * 1e3, 4),
                        "frac_of_value": round((total_segments / dt3a * 1e-6) / value, 4) if value > 0 else None,
                        "note": "drt_hip_render_async + drt_hip_wait with host out_rgb / out_param_grad, up to four frames in flight "
                                "(PCIe D2H of image and gradients included, overlapped with the next frame's kernels)",
                        "sync": {"value": round(total_segments / dt3p * 1e-6, 2), "ms_per_step": round(dt3p * 1e3, 4),
                                 "note": "drt_hip_render: returns with the results in the caller's buffers; out_rgb pinned once with "
                                         "drt_hip_pin_host (the finishing kernel stores the image straight into it; the wait polls a completion word)",
                                 "pageable": {"value": round(total_segments / dt3 * 1e-6, 2), "ms_per_step": round(dt3 * 1e3, 4),
                                              "note": "the same call with out_rgb in pageable memory: image stored into the context's pinned "
                                                      "block, then a 3 MB memcpy on the host"}}}

    # N > 1: efficiency against the stored single-GPU value of the SAME per-GPU workload (profiles/n1_reference.json, written by
    # a 1-GPU run with --store-n1; the driver computes its own figure from its own N = 1 run)
    weak_scaling = None
    n1_path = os.path.join(ROOT, "profiles", "n1_reference.json")
    if rank == 0:
        try:
            n1 = json.load(open(n1_path)) if os.path.exists(n1_path) else {}
        except Exception:
            n1 = {}
        if world == 1 and a.store_n1 and not use_dist:
            n1[workload_key] = {"value": round(value, 2), "unit": "Mray/s", "ms_per_step": round(ms_per_step, 4),
                                "serial_value": serial_view["value"] if serial_view else None}
            json.dump(n1, open(n1_path, "w"), indent=1, sort_keys=True)
        if world > 1 and workload_key in n1:
            ref1 = n1[workload_key]["value"]
            weak_scaling = {"efficiency": round(value / (world * ref1), 4), "n1_value": ref1, "unit": "Mray/s",
                            "note": f"value / ({world} x the stored 1-GPU value of this per-GPU workload, profiles/n1_reference.json)"}
            # (the same against frames in stream order -- what an optimisation loop gets --, where both numbers exist)
            if serial_view and n1[workload_key].get("serial_value"):
                weak_scaling["efficiency_serial_frames"] = round(serial_view["value"] / (world * n1[workload_key]["serial_value"]), 4)
        elif world > 1:
            weak_scaling = {"efficiency": None, "note": f"no stored 1-GPU value for '{workload_key}' (run bench.py --store-n1 on one GPU)"}

    if rank == 0:
        what = "fwd+bwd" if backward else "fwd"
        if world > 1:
            par = (f"{dist.get_world_size()} ranks x 1 GPU ({'; '.join(devices)}): interleaved 16-row bands, ONE "
                   f"{('ncclAllReduce(sum, f64, P x 3) inside libdrt_hip.so' + (' on its second stream (overlaps the next step)' if a.allreduce == 'async' else '')) if reduce_mode == 'library' else 'torch.distributed all_reduce'}"
                   f" per step" + (" [ranks share device 0: plumbing test, numbers meaningless]" if a.same_gpu else ""))
        elif use_dist:
            par = f"1 rank, 1 GPU ({devices[0]}); gradient through the {reduce_mode or 'no'} all-reduce of a 1-rank communicator"
        else:
            par = "1 GPU"
        line = {
            "metric": f"Mray/s ({what}), {'Cornell' if a.scene.startswith('cornell') else a.scene} {a.width}x{a.height} @{a.spp}spp {depth_text}",
            "value": round(value, 2), "unit": "Mray/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.config_name}; scene '{a.scene}' (cornell box of render.cpp:26-59"
                                   f"{' + procedural mesh' if a.scene.startswith('mesh') else ''}) {a.width}x{a.height}, "
                                   f"{a.spp} spp per GPU, {depth_text} (-b {a.min_bounces} -p {a.absorb:g}), "
                                   f"{'fwd + radiative-backprop gradients of ' + str(scene.n_params) + ' parameters' if backward else 'fwd only'}",
                       "mode": (("2 frames pipelined (the path kernels of consecutive frames overlap on two streams); "
                                 if (stats.get("path_program", "none") != "none" and os.environ.get("DRT_HIP_OVERLAP_FRAMES", "1") != "0")
                                 else "frames in stream order; ") +
                                {"builtin": "k_path with the shape kinds of the reference's scene compiled in (the instantiation the library carries)",
                                 "specialised": "k_path compiled for this scene's shape kinds at run time (hiprtc)",
                                 "sorted": "k_path reading the shape kinds at run time (kind-sorted program)",
                                 "none": "the queue wavefront (K1-K7)"}.get(stats.get("path_program", "none"), "?")),
                       "program": stats.get("path_program"), "specialise_ms": round(stats.get("jit_ms", 0.0), 1),
                       "paths_per_step": int(total_paths), "rays_per_step": int(total_segments),
                       "parallelism": par, "batches_per_step": stats["batches"],
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)") +
                                            (" (set by this script before the HIP runtime started)" if "--same-gpu" not in sys.argv else ""),
                       "capped_paths_per_step": stats["capped_paths"]},
            "weak_scaling": weak_scaling,
            "serial_frame": serial_view, "generic_program": generic_view,
            "roofline": roofline, "cpu_baseline": cpu_baseline, "host_buffers": host_buffers,
            "preheat_ms": a.preheat_ms, "preheat_frames": preheat_frames,
            "f64": f64_view, "fwd_only": fwd_view, "unbiased": unb_view, "two_contexts": two_ctx_view, "many_parameters": many_view,
        }
    # teardown in the same order on every rank: the library's communicator first (all ranks are still here), then the
    # launcher's process group, then the context
    r.synchronize()
    if use_dist:
        if reduce_mode == "library":
            r.comm_destroy()
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    if rank == 0:
        # the ONE line, and the LAST one of this rank's stdout: RCCL writes its version banner through C stdio, which keeps it
        # in a buffer until the process ends when stdout is a pipe -- flushed here, before the line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
