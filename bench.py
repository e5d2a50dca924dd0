#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Metric: Mray/s (fwd+bwd) on the Cornell box of /root/reference/src/render.cpp:26-59,
512x512, 64 spp, depth 8 (`-b 8 -p 1`), gradients w.r.t. the material albedos + emission
(BASELINE config 3).  A "step" is one full render call: every path of the frame traced forward
through the wavefront pipeline, the tape swept backward, gradients reduced.  ray = one raycast
(camera ray included), exactly what the reference's Pathtracer::raycast counts.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Multi-GPU: one process per GPU; the frame's rows are dealt to the ranks in interleaved bands
and spp is multiplied by N, so every rank traces the same 512*512*64 paths whatever N is (weak
scaling); the only collective is one RCCL all-reduce of the P x 3 gradient vector per step.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# vector-instruction issue: a wave64 VALU op takes 2 cycles of its SIMD; 256 CUs x 4 SIMDs at 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0     # 1228.8 G wave-instructions / s
# algorithmic bytes per ray segment, f32 queues
# shade (unfused): ray 24 + id 8 + hit 8 read; ray 24 + id 8 + tape 8 written per segment.
# Fused shade (K2 folded into K3, several bounces per launch in registers): 8 B of tape per segment + 32 B
# per ray a launch READS from the queue + 32 B per survivor it WRITES back -- the library counts both.
BYTES_PER_UNIT = {"intersect": 32.0, "shade": 80.0, "backward": 8.0}


def _oracle_shard(args):
    """Worker of the all-cores CPU figure: one row-band shard of the sample, in its own process."""
    scene_name, w, h, spp, depth, backward, shard, n_shards = args
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    rp = pkg.RenderParams(spp=spp, min_bounces=depth, absorb=1.0, seed=1, shard=shard, n_shards=n_shards, band_rows=4)
    r = oracle.render(pkg.scene_by_name(scene_name), pkg.cornell_camera(w, h), rp, backward=backward)
    return r["stats"]["segments"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--scene", default="cornell", help="cornell (headline) | cornell_specular | mesh<lat>x<lon>[f<n>] | random<seed>")
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--unbiased", action="store_true", help="backward with the unbiased integration operator")
    ap.add_argument("--batch-paths", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-streaming-view", action="store_true",
                    help="skip the extra one-launch-per-bounce measurement (profiling runs: keeps kernel statistics unmixed)")
    ap.add_argument("--bounces-per-launch", type=int, default=0, help="0 = automatic (drt_hip.h)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the real thing) | gloo: lets two ranks share ONE GPU to exercise the N > 1 path "
                         "on a single-GPU box (with --same-gpu); the numbers of such a run mean nothing")
    ap.add_argument("--same-gpu", action="store_true", help="every rank uses device 0 (testing only)")
    ap.add_argument("--cpu-spp", type=int, default=48)
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="also time the oracle on every host core (independent row-band processes)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, a.gpus):
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # launched by torch.distributed.run (also with one rank: exercises the RCCL path on one GPU)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if a.same_gpu:
        local_rank = 0
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.dist_backend)

    pkg = entry.load_package()
    scene = pkg.scene_by_name(a.scene)
    cam = pkg.cornell_camera(a.width, a.height)
    backward = not a.forward_only
    rp = pkg.RenderParams(spp=a.spp * world, min_bounces=a.depth, absorb=1.0, seed=1,
                          shard=rank, n_shards=world, band_rows=16, batch_paths=a.batch_paths,
                          flags=pkg.RENDER_UNBIASED if (a.unbiased and backward) else 0,
                          bounces_per_launch=a.bounces_per_launch)

    r = pkg.HipRenderer(local_rank)          # raises without libdrt_hip.so / a device: no fallback
    r.upload_scene(scene)
    dev = torch.device("cuda", local_rank)
    out_rgb = torch.zeros((a.height, a.width, 3), dtype=torch.float32, device=dev)
    # two gradient buffers: the all-reduce of step i runs on its own stream while step i + 1 renders
    # into the other buffer (every step's gradient is complete and reduced; only its delivery overlaps
    # the next step's compute -- "collectives on a separate stream", the usual data-parallel overlap)
    grads = [torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in range(2)]
    ext = torch.cuda.ExternalStream(r.stream, device=dev)
    comm = torch.cuda.Stream(device=dev) if use_dist else None
    reduced = [None, None]          # event: the all-reduce that last used this buffer has finished
    n_step = [0]

    def step(timing=False):
        b = n_step[0] & 1
        n_step[0] += 1
        grad = grads[b]
        if reduced[b] is not None:
            ext.wait_event(reduced[b])         # step i - 2's all-reduce read this buffer
        st = r.render_device(cam, rp, out_rgb.data_ptr(), grad.data_ptr() if backward else 0,
                             backward=backward, timing=timing, sync=False)
        if use_dist and backward:
            rendered = torch.cuda.Event()
            rendered.record(ext)
            comm.wait_event(rendered)
            with torch.cuda.stream(comm):
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)   # the ONE collective of the path
                done = torch.cuda.Event()
                done.record(comm)
            reduced[b] = done
        return st

    def fence():
        r.synchronize()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel HIP-event timing (events on the context's own stream), same workload, same
    # process, right after the timed region; also yields the segment count of one step
    kernel_ms = {k: 0.0 for k in pkg.KERNEL_NAMES}
    kernel_launches = {k: 0 for k in pkg.KERNEL_NAMES}
    n_prof = max(1, min(a.steps, 5))
    stats = None
    for _ in range(n_prof):
        stats = step(timing=True)
        for k in pkg.KERNEL_NAMES:
            kernel_ms[k] += stats["kernels"][k]["ms"]
            kernel_launches[k] += stats["kernels"][k]["launches"]
    fence()
    segments = stats["segments"]
    paths = stats["paths"]
    queue_rays = stats["queue_rays_read"] + stats["queue_rays_written"]
    if use_dist:
        t = torch.tensor([segments, paths], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_segments, total_paths = float(t[0].item()), float(t[1].item())
    else:
        total_segments, total_paths = float(segments), float(paths)

    ms_per_step = elapsed / a.steps * 1e3
    value = total_segments * a.steps / elapsed * 1e-6   # Mray/s, whole job

    # dominant kernel and its roofline (rank 0's device; every rank runs the same work)
    units = {"intersect": segments, "shade": segments, "backward": segments,
             "raygen": paths, "film": paths, "gradreduce": 0}
    bpu = {"intersect": BYTES_PER_UNIT["intersect"],
           # fused shade: tape 8 B/segment + queued rays; when it also walks the tape in place (forward-only, one launch
           # from the eye to the path's end: no K6 / k_radiance launch) the tape stays in LDS and 16 B of radiance
           # per path is all it writes
           "shade": BYTES_PER_UNIT["shade"] if kernel_launches["intersect"] else
                    (8.0 if kernel_launches["backward"] else 16.0 * paths / max(1, segments)) + 32.0 * queue_rays / max(1, segments),
           "backward": BYTES_PER_UNIT["backward"], "raygen": 32.0, "film": 16.0, "gradreduce": 0.0}
    per_kernel = {}
    for k in pkg.KERNEL_NAMES:
        ms = kernel_ms[k] / n_prof
        if kernel_launches[k] == 0:
            continue
        gbs = units[k] * bpu[k] / (ms * 1e-3) * 1e-9 if ms > 0 else 0.0
        per_kernel[k] = {"ms_per_step": round(ms, 4), "launches_per_step": kernel_launches[k] // n_prof,
                         "bytes_per_unit": round(bpu[k], 3), "achieved_GBs": round(gbs, 1),
                         "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    dominant = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"]) if per_kernel else "shade"
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    # the committed rocprofv3 counters were taken on the headline workload: only quote them for it
    pmc_applies = (a.scene == "cornell" and (a.width, a.height, a.spp, a.depth) == (512, 512, 64, 8) and backward
                   and not a.unbiased and a.bounces_per_launch == 0)
    if pmc_applies and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get(dominant)
            traffic = tj.get("GBs") if tj else None
        except Exception:
            traffic = None
    dk = per_kernel.get(dominant, {"achieved_GBs": 0.0, "launches_per_step": 1, "ms_per_step": 0.0})
    roofline = {"bound": "hbm", "kernel": "k_" + dominant, "achieved": dk["achieved_GBs"],
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dk["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_note": "GB/s of PMC-counted HBM bytes per launch (profiles/traffic.json) over the profiled launch time" if traffic else None,
                "avg_launch_ms": round(dk["ms_per_step"] / max(1, dk["launches_per_step"]), 4),
                # the kernel that finds the closest hit: k_intersect(+mesh), or k_shade when K2 is folded into it
                "traversal_kernel": dict(per_kernel.get("intersect") or per_kernel.get("shade") or {},
                                         name="k_intersect" if "intersect" in per_kernel else "k_shade<fused>"),
                "kernels": per_kernel}
    shade = per_kernel.get("shade")
    multi_bounce = (shade is not None and not kernel_launches["intersect"]
                    and shade["launches_per_step"] < a.depth * max(1, stats["batches"]) and not a.unbiased)
    if multi_bounce:
        # The fused shade kernel keeps a ray in registers over several bounces: the queue traffic is paid once
        # per launch, not per segment, and the kernel is bound by vector-instruction issue, not by HBM.  Two
        # extra views so the HBM fraction above is not read alone:
        #  (1) the vector-issue utilisation, from the rocprofv3 SQ_INSTS_VALU count of the same kernel;
        #  (2) the SAME workload with one launch per bounce (bounces_per_launch = 1, the streaming wavefront
        #      the HBM roofline describes), timed live here.
        roofline["note"] = (f"k_shade runs {a.depth * max(1, stats['batches']) // max(1, shade['launches_per_step'])} bounces per launch in registers "
                            f"({shade['bytes_per_unit']} B/segment): VALU-issue-bound; see roofline.valu and roofline.streaming")
        try:
            tsh = json.load(open(tpath)).get("shade", {}) if pmc_applies else {}
        except Exception:
            tsh = {}
        if tsh.get("valu_insts_per_launch"):
            ginst = tsh["valu_insts_per_launch"] / (shade["ms_per_step"] / shade["launches_per_step"] * 1e-3) * 1e-9
            roofline["valu"] = {"achieved": round(ginst, 1), "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
                                "frac": round(ginst / VALU_PEAK_GINST, 4),
                                "note": "SQ_INSTS_VALU per launch (profiles/traffic.json, rocprofv3 --pmc) / live launch time; peak = one "
                                        "VALU op per 2 cycles per SIMD (two interleaved waves); the CU's one scalar unit is about as busy "
                                        "(DESIGN.md section 3)"}
    if multi_bounce and not a.no_streaming_view:
        import dataclasses
        rp1 = dataclasses.replace(rp, bounces_per_launch=1)

        def step1(timing=False):
            return r.render_device(cam, rp1, out_rgb.data_ptr(), grads[0].data_ptr() if backward else 0,
                                   backward=backward, timing=timing, sync=False)
        for _ in range(2):
            step1()
        fence()
        t4 = time.perf_counter()
        for _ in range(n_prof):
            step1()
        fence()
        dt4 = (time.perf_counter() - t4) / n_prof
        ms1, q1 = 0.0, 0
        for _ in range(n_prof):
            st1 = step1(timing=True)
            ms1 += st1["kernels"]["shade"]["ms"]
            q1 = st1["queue_rays_read"] + st1["queue_rays_written"]
        fence()
        ms1 /= n_prof
        bpu1 = 8.0 + 32.0 * q1 / max(1, segments)
        gbs1 = segments * bpu1 / (ms1 * 1e-3) * 1e-9 if ms1 > 0 else 0.0
        roofline["traversal_kernel"]["one_launch_per_bounce"] = {"ms_per_step": round(ms1, 4), "achieved_GBs": round(gbs1, 1),
                                                                 "frac_hbm_peak": round(gbs1 / HBM_PEAK_GBS, 4)}
        roofline["streaming"] = {"bounces_per_launch": 1, "value": round(segments / dt4 * 1e-6, 2), "unit": "Mray/s",
                                 "ms_per_step": round(dt4 * 1e3, 4), "kernel": "k_shade", "bytes_per_unit": round(bpu1, 3),
                                 "achieved": round(gbs1, 1), "peak": HBM_PEAK_GBS, "frac": round(gbs1 / HBM_PEAK_GBS, 4)}

    cpu_baseline = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        oracle = entry.load_oracle()
        crp = pkg.RenderParams(spp=a.cpu_spp, min_bounces=a.depth, absorb=1.0, seed=1)
        t1 = time.perf_counter()
        ref = oracle.render(scene, cam, crp, backward=backward)
        dt = time.perf_counter() - t1
        cpu_baseline = {"value": round(ref["stats"]["segments"] / dt * 1e-6, 3), "unit": "Mray/s",
                        "cores": 1, "kind": "port",
                        "sample": f"same scene and frame {a.width}x{a.height}, {a.cpu_spp} of the {a.spp} spp, "
                                  f"depth {a.depth}, {'fwd+bwd' if backward else 'fwd'}: "
                                  f"{ref['stats']['segments']} rays in {dt:.1f} s (fp64 C restatement, 1 thread)"}
        # parity of this very workload's gradients against the CPU restatement (same RNG keys):
        # the first cpu_spp samples of every pixel are the same paths on both sides only when
        # spp matches, so run the device once more at the sample's spp
        if backward:
            img_d, g_d, _ = r.render(cam, crp, backward=True)
            gerr = float(np.abs(g_d - ref["grads"]).max() / np.abs(ref["grads"]).max())
            cpu_baseline["grad_max_rel_err_vs_cpu"] = gerr
        if a.cpu_all_cores:
            # the reference is single-threaded by construction (global rand()); this is N independent
            # processes, each rendering its interleaved row bands of the same sample
            import multiprocessing as mp
            n = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
            jobs = [(a.scene, a.width, a.height, a.cpu_spp, a.depth, backward, i, n) for i in range(n)]
            with mp.get_context("spawn").Pool(n) as pool:
                # start the workers and load the checker in each of them before the clock starts
                pool.map(_oracle_shard, [(a.scene, 16, 16, 1, 1, False, 0, 1)] * n, chunksize=1)
                t2 = time.perf_counter()
                segs = sum(pool.map(_oracle_shard, jobs, chunksize=1))
                dt2 = time.perf_counter() - t2
            cpu_baseline["all_cores"] = {"value": round(segs / dt2 * 1e-6, 2), "unit": "Mray/s", "cores": n,
                                         "note": "N independent row-band processes of the same sample, workers warmed up; N = usable CPUs reported by the OS (a container CPU quota may be lower)"}

    # the same call through the host-buffer entry point (synchronous, image and gradients copied to
    # pageable host memory over PCIe): reported beside `value`, never as `value`
    host_buffers = None
    if rank == 0 and world == 1:
        hrp = pkg.RenderParams(spp=a.spp, min_bounces=a.depth, absorb=1.0, seed=1, batch_paths=a.batch_paths)
        n_host = max(1, min(a.steps, 5))
        r.render(cam, hrp, backward=backward, unbiased=a.unbiased and backward)
        t3 = time.perf_counter()
        for _ in range(n_host):
            r.render(cam, hrp, backward=backward, unbiased=a.unbiased and backward)
        dt3 = (time.perf_counter() - t3) / n_host
        host_buffers = {"value": round(total_segments / dt3 * 1e-6, 2), "unit": "Mray/s",
                        "ms_per_step": round(dt3 * 1e3, 4),
                        "note": "drt_hip_render with host out_rgb / out_param_grad (synchronous, PCIe D2H of the image included)"}

    if rank == 0:
        line = {
            "metric": f"Mray/s ({'fwd+bwd' if backward else 'fwd'}), Cornell {a.width}x{a.height} @{a.spp}spp depth {a.depth}",
            "value": round(value, 2), "unit": "Mray/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("" if a.scene == "cornell" else f"[scene {a.scene}] ") + f"cornell box (render.cpp:26-59) {a.width}x{a.height}, "
                                   f"{a.spp} spp per GPU, depth {a.depth} (-b {a.depth} -p 1), "
                                   f"{'fwd + radiative-backprop grads of 4 params' if backward else 'fwd only'}",
                       "paths_per_step": int(total_paths), "rays_per_step": int(total_segments),
                       "parallelism": f"pixel-row bands x{world}, 1 grad all-reduce" if world > 1 else "1 GPU",
                       "batches_per_step": stats["batches"]},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "host_buffers": host_buffers,
        }
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
