"""Pixel sharding across the GPUs of one node: one process per GPU.  Every (pixel, sample) path is
independent (/root/reference/src/render.cpp:72-81); the only cross-rank state is the P x 3
parameter-gradient accumulator (vector.hpp:185-188), summed with ONE all-reduce per render.
Image rows are disjoint per rank and need no collective.

The all-reduce itself lives behind the C ABI (libdrt_hip.so links RCCL: drt_hip_comm_init_rank +
DRT_RENDER_ALLREDUCE); the launcher's job is only to hand rank 0's 128-byte communicator id to the
other ranks -- join_library_communicator() does that over whatever torch.distributed group the job
already has (nccl or gloo).  allreduce_grads() is the same sum done by torch.distributed: used by the
CPU tests (gloo, the oracle standing in for the device) and as bench.py's --reduce torch A/B."""
from __future__ import annotations

import dataclasses
from typing import Callable, Optional

import numpy as np


def shard_params(rp, rank: int, world: int, band_rows: Optional[int] = None):
    """RenderParams of `rank`: interleaved row bands so depth imbalance averages out."""
    return dataclasses.replace(rp, shard=rank if world > 1 else 0, n_shards=max(1, world),
                               band_rows=band_rows or rp.band_rows)


def join_library_communicator(renderer, pkg, group=None) -> bool:
    """Give `renderer` (a HipRenderer of this rank) its rank in a communicator spanning the ranks of `group`.
    Collective.  -> True when every rank joined (renders may then carry RENDER_ALLREDUCE), False -- and no rank
    keeps a communicator -- when any of them could not (e.g. two test ranks on one device).

    ncclCommInitRank blocks until EVERY rank has entered it, so nobody may enter before all ranks are known to be
    able to: rank 0's id is broadcast (None when it could not be made), every rank checks its own preconditions (a
    plain context without a communicator) and the ranks agree (MIN) before the collective call."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    uid = [None]
    if rank == 0:
        try:
            uid = [pkg.comm_unique_id()]
        except Exception:
            uid = [None]
    dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ready = uid[0] is not None and renderer.group_size == 1 and renderer.comm_size == 0
    ok = torch.tensor([1 if ready else 0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        return False
    joined = True
    try:
        renderer.comm_init(uid[0], rank, world)
    except pkg.DrtHipError:
        joined = False
    ok = torch.tensor([1 if joined else 0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0 and joined:
        renderer.comm_destroy()
    return int(ok.item()) == 1


def allreduce_grads(grads, group=None):
    """SUM all-reduce of the parameter gradients (numpy [P,3] float64, or a torch tensor that
    already lives on this rank's device). No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    if isinstance(grads, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(grads))
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()
    dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    return grads


def render_sharded(render_fn: Callable, rp, rank: int, world: int, group=None):
    """render_fn(rp_shard) -> (image [H,W,3] with only this shard's rows written, grads [P,3]).
    Returns (image of this shard, all-reduced grads)."""
    img, grads = render_fn(shard_params(rp, rank, world))
    if grads is not None:
        grads = allreduce_grads(grads, group)
    return img, grads


def gather_image(img: np.ndarray, group=None) -> np.ndarray:
    """Assemble the full frame on every rank (rows are disjoint and zero elsewhere: a sum).
    Only for writing the result out; not part of the timed path."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return img
    t = torch.from_numpy(np.ascontiguousarray(img))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()
