"""The N > 1 path on CPU: two gloo ranks, each renders its interleaved row bands (the oracle
stands in for the device here -- tests may use it), ONE sum all-reduce of the gradient vector,
disjoint image rows.  Result must equal the single-process render."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import importlib.util
    import torch.distributed as dist
    import __graft_entry__ as entry
    pkg = entry.load_package()
    oracle = entry.load_oracle()
    spec = importlib.util.spec_from_file_location(
        "drt_distributed", os.path.join(ROOT, "differentiable-renderer_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(40, 36)
    rp = pkg.RenderParams(spp=4, min_bounces=2, absorb=0.4, seed=6, band_rows=4)

    def render_fn(rps):
        r = oracle.render(scene, cam, rps, backward=True)
        return r["image"], r["grads"]

    img, grads = D.render_sharded(render_fn, rp, rank, world)
    rows = pkg.shard_rows(cam.height, 4, world, rank)
    other = np.setdiff1d(np.arange(cam.height), rows)
    assert not img[other].any()
    full = D.gather_image(img)
    np.save(os.path.join(out_dir, f"img{rank}.npy"), full)
    np.save(os.path.join(out_dir, f"grad{rank}.npy"), grads)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_equal_one(pkg, oracle, tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(40, 36)
    rp = pkg.RenderParams(spp=4, min_bounces=2, absorb=0.4, seed=6, band_rows=4)
    ref = oracle.render(scene, cam, rp, backward=True)
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"img{r}.npy"), ref["image"])
        np.testing.assert_allclose(np.load(tmp_path / f"grad{r}.npy"), ref["grads"], rtol=1e-12)
