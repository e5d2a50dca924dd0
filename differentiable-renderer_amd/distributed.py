"""Pixel sharding across the GPUs of one node: one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).  Every (pixel, sample) path is
independent (/root/reference/src/render.cpp:72-81); the only cross-rank state is the P x 3
parameter-gradient accumulator (vector.hpp:185-188), summed with ONE all-reduce per render.
Image rows are disjoint per rank and need no collective."""
from __future__ import annotations

import dataclasses
from typing import Callable, Optional

import numpy as np


def shard_params(rp, rank: int, world: int, band_rows: Optional[int] = None):
    """RenderParams of `rank`: interleaved row bands so depth imbalance averages out."""
    return dataclasses.replace(rp, shard=rank if world > 1 else 0, n_shards=max(1, world),
                               band_rows=band_rows or rp.band_rows)


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
