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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_n_ranks_equal_one(pkg, oracle, tmp_path, world):
    """2 and 8 ranks (the node size of BASELINE configs 4 and 5): every rank's bands, one sum all-reduce."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(40, 36)
    rp = pkg.RenderParams(spp=4, min_bounces=2, absorb=0.4, seed=6, band_rows=4)
    ref = oracle.render(scene, cam, rp, backward=True)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / f"img{r}.npy"), ref["image"])
        np.testing.assert_allclose(np.load(tmp_path / f"grad{r}.npy"), ref["grads"], rtol=1e-12)


class _StubRenderer:
    """What join_library_communicator touches of a HipRenderer (no device needed)."""
    def __init__(self, comm_size=0, group_size=1, fail_init=False):
        self.comm_size, self.group_size, self.fail_init, self.inits, self.destroys = comm_size, group_size, fail_init, 0, 0

    def comm_init(self, uid, rank, world):
        self.inits += 1
        if self.fail_init:
            raise _StubPkg.DrtHipError("ncclCommInitRank failed")
        self.comm_size = world

    def comm_destroy(self):
        self.destroys += 1
        self.comm_size = 0


class _StubPkg:
    class DrtHipError(RuntimeError):
        pass
    uid_fails = False

    @classmethod
    def comm_unique_id(cls):
        if cls.uid_fails:
            raise cls.DrtHipError("no id")
        return b"x" * 128


def _join_worker(rank, world, port, out_dir, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location(
        "drt_distributed", os.path.join(ROOT, "differentiable-renderer_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = _StubRenderer(comm_size=1 if (case == "busy" and rank == 1) else 0, fail_init=(case == "init_fails" and rank == 1))
    _StubPkg.uid_fails = case == "no_uid"
    ok = D.join_library_communicator(r, _StubPkg)
    np.save(os.path.join(out_dir, f"join{rank}.npy"), np.array([int(ok), r.inits, r.destroys, r.comm_size]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("case", ["fine", "busy", "no_uid", "init_fails"])
def test_joining_the_library_communicator_never_strands_a_rank(tmp_path, case):
    """ncclCommInitRank blocks until every rank has entered it: a rank that cannot join (it already has a communicator,
    rank 0 could not make the id) must be found out BEFORE anyone enters -- nobody calls comm_init then; a failure inside
    the collective call leaves no rank with a communicator."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_join_worker, args=(3, port, str(tmp_path), case), nprocs=3, join=True)
    res = [np.load(tmp_path / f"join{r}.npy") for r in range(3)]
    if case == "fine":
        assert all(r[0] == 1 and r[1] == 1 and r[3] == 3 for r in res)
    elif case in ("busy", "no_uid"):
        assert all(r[0] == 0 and r[1] == 0 for r in res)          # nobody entered the collective
    else:
        assert all(r[0] == 0 and r[1] == 1 for r in res)
        assert res[0][2] == 1 and res[0][3] == 0 and res[2][2] == 1   # the ranks that had joined let go again
