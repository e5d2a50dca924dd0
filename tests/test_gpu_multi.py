"""The multi-GPU surface of the C ABI on a single-GPU box: group contexts (one process, n devices;
drt_hip_create_group) and per-rank communicators (one process per GPU; drt_hip_comm_init_rank).  The
gradient reduction -- THE collective of the path, VariableNode::backward's `m_grad += grad`
(/root/reference/include/drt/vector.hpp:185-188) summed across devices -- runs inside libdrt_hip.so:
same-device members are added on the device, distinct devices by one ncclAllReduce (here a 1-rank
communicator: the call is made, the code path is the one 8 GPUs take)."""
import dataclasses
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_group_context_equals_one_context(pkg, hip):
    """A group that lists device 0 three times: three members deal the row bands among themselves, the library sums
    their gradients (on-device adds + the all-reduce of the leaders' communicator) and assembles the frame."""
    scene = pkg.cornell_box(front_specular=True)
    cam = pkg.cornell_camera(96, 70)
    rp = pkg.RenderParams(spp=6, min_bounces=3, absorb=0.3, seed=4, band_rows=8)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True)
    g = pkg.HipRenderer([0, 0, 0])
    try:
        assert g.group_size == 3 and g.comm_size == 1
        g.upload_scene(scene)
        gimg, ggrads, gst = g.render(cam, rp, backward=True)
        np.testing.assert_array_equal(gimg, img)                  # disjoint rows, same kernels
        np.testing.assert_allclose(ggrads, grads, rtol=1e-9)      # same terms, summed in another order
        assert gst["segments"] == st["segments"] and gst["paths"] == st["paths"]
        # update_params reaches every member
        newp = np.array(scene.params) * 0.5 + 0.1
        g.update_params(newp)
        hip.update_params(newp)
        a = g.render(cam, rp, backward=True)
        b = hip.render(cam, rp, backward=True)
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_allclose(a[1], b[1], rtol=1e-9)
        # forward only, and the per-pixel gradient image, through the group
        f = g.render(cam, rp, backward=False)
        np.testing.assert_array_equal(f[0], b[0])
        gi_g = g.render_gradient_image(cam, rp, 2)
        gi_1 = hip.render_gradient_image(cam, rp, 2)
        np.testing.assert_array_equal(gi_g[1], gi_1[1])
        # a group returns through host buffers only
        with pytest.raises(pkg.DrtHipError, match="DRT_ERR_UNSUPPORTED"):
            g.render_device(cam, rp, 0, 0)
    finally:
        g.close()


def test_group_members_launch_side_by_side_on_the_blocking_routes(pkg, hip):
    """The launches that can block the host -- the queue route of a roulette-terminated render on a mesh scene asks the
    device every few bounces whether a path is left, a pageable adjoint image is copied synchronously -- are issued from
    one host thread per member (and, with DRT_HIP_GROUP_THREADS=0 in a fresh process, one after the other): same frame,
    same gradients, same counts as one context."""
    scene = pkg.scene_by_name("mesh12x16")
    cam = pkg.cornell_camera(72, 64)
    rp = pkg.RenderParams(spp=5, min_bounces=2, absorb=0.25, seed=21, band_rows=8)
    adj = np.random.RandomState(5).uniform(0.2, 1.5, (64, 72, 3)).astype(np.float32)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True, adjoint=adj)
    g = pkg.HipRenderer([0, 0, 0, 0])
    try:
        g.upload_scene(scene)
        for _ in range(3):                                        # (threads are started per call)
            gimg, ggrads, gst = g.render(cam, rp, backward=True, adjoint=adj)
            np.testing.assert_array_equal(gimg, img)
            np.testing.assert_allclose(ggrads, grads, rtol=1e-9)
            assert gst["segments"] == st["segments"] and gst["paths"] == st["paths"]
    finally:
        g.close()
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import __graft_entry__ as e; pkg = e.load_package();"
            "scene = pkg.scene_by_name('mesh12x16'); cam = pkg.cornell_camera(72, 64);"
            "rp = pkg.RenderParams(spp=5, min_bounces=2, absorb=0.25, seed=21, band_rows=8);"
            "adj = np.random.RandomState(5).uniform(0.2, 1.5, (64, 72, 3)).astype(np.float32);"
            "g = pkg.HipRenderer([0, 0, 0, 0]); g.upload_scene(scene); im, gr, st = g.render(cam, rp, backward=True, adjoint=adj);"
            "print('RESULT', float(im.sum()), repr(gr.ravel().tolist()), st['segments'])" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DRT_HIP_GROUP_THREADS="0"), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):]   # (RCCL prints a banner at exit)
    tot = float(line.split(" ", 1)[0])
    segs = int(line.rsplit(" ", 1)[1])
    gr = np.array(json.loads(line.split(" ", 1)[1].rsplit(" ", 1)[0]))
    assert tot == float(img.sum()) and segs == st["segments"]
    np.testing.assert_allclose(gr, grads.ravel(), rtol=1e-9)


def test_group_as_one_shard_of_a_larger_job(pkg, hip):
    """rp.shard / n_shards address a GROUP as one node of a multi-node job: 2 'nodes' x 2 members tile the frame."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(64, 64)
    rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=9, band_rows=4)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True)
    g = pkg.HipRenderer([0, 0])
    try:
        g.upload_scene(scene)
        acc, gsum = np.zeros_like(img), np.zeros_like(grads)
        for node in range(2):
            im, gr, _ = g.render(cam, dataclasses.replace(rp, shard=node, n_shards=2), backward=True)
            rows = np.concatenate([pkg.shard_rows(64, 4, 4, node * 2 + m) for m in range(2)])
            other = np.setdiff1d(np.arange(64), rows)
            assert not im[other].any()
            acc += im
            gsum += gr
        np.testing.assert_array_equal(acc, img)
        np.testing.assert_allclose(gsum, grads, rtol=1e-9)
    finally:
        g.close()


def test_in_library_allreduce_with_a_one_rank_communicator(pkg):
    """One process per GPU: unique id -> comm_init_rank -> DRT_RENDER_ALLREDUCE.  With one rank the sum is the
    rank's own gradient; the ncclAllReduce is enqueued on the context's stream like on 8 GPUs."""
    r = pkg.HipRenderer(0)
    try:
        scene = pkg.cornell_box()
        cam = pkg.cornell_camera(48, 48)
        rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=3)
        r.upload_scene(scene)
        _, g0, _ = r.render(cam, rp, backward=True)
        with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):      # no communicator yet
            r.render(cam, dataclasses.replace(rp, flags=pkg.RENDER_ALLREDUCE), backward=True)
        assert r.comm_size == 0
        r.comm_init(pkg.comm_unique_id(), 0, 1)
        assert r.comm_size == 1
        _, g1, _ = r.render(cam, dataclasses.replace(rp, flags=pkg.RENDER_ALLREDUCE), backward=True)
        np.testing.assert_array_equal(g1, g0)
        with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):      # one communicator per context
            r.comm_init(pkg.comm_unique_id(), 0, 1)
        r.comm_destroy()
        assert r.comm_size == 0
    finally:
        r.close()


def _bench(*args, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=timeout, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` must itself start two ranks (here sharing device 0 over a gloo rendezvous:
    plumbing only) and report n_gpus == 2; the numbers of such a run mean nothing."""
    line = _bench("--gpus", "2", "--dist-backend", "gloo", "--same-gpu", "--steps", "2", "--warmup", "1",
                  "--width", "128", "--height", "128", "--spp", "4", "--no-cpu-baseline", "--no-extra-views")
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert "2 ranks" in line["config"]["parallelism"]
    assert line["config"]["paths_per_step"] == 2 * 128 * 128 * 4      # weak scaling: per-rank work is fixed


@pytest.mark.timeout(900)
def test_bench_single_rank_under_torchrun_uses_the_library_collective():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                          "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--width", "128", "--height", "128", "--spp", "4",
                          "--no-cpu-baseline", "--no-extra-views"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and "library all-reduce" in line["config"]["parallelism"]


def test_async_host_renders_match_the_synchronous_call_bit_for_bit(pkg, hip):
    """drt_hip_render_async / drt_hip_wait: up to four frames in flight, the copies of frame i overlap the next frames' kernels;
    images, gradients and segment counts equal drt_hip_render's exactly; the documented refusals hold."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(96, 64)
    hip.upload_scene(scene)
    rps = [pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=s) for s in range(1, 12)]
    ref = [hip.render(cam, rp, backward=True) for rp in rps]
    for depth in (2, 3, pkg.FRAMES_IN_FLIGHT):             # frames kept in flight (from three on their path kernels overlap)
        handles, got = [], []
        for i, rp in enumerate(rps):
            handles.append(hip.render_async(cam, rp, backward=True))
            if len(handles) == depth:                      # frame i is enqueued, frame i - depth + 1 is collected
                got.append(hip.wait(handles.pop(0)))
        while handles:
            got.append(hip.wait(handles.pop(0)))
        for (img, grads, st), (rimg, rgrads, rst) in zip(got, ref):
            np.testing.assert_array_equal(img, rimg)
            np.testing.assert_array_equal(grads, rgrads)
            assert st["segments"] == rst["segments"] and st["paths"] == rst["paths"]
    # one frame more than FRAMES_IN_FLIGHT, a synchronous render while frames are in flight, an unknown ticket: refused
    more = [hip.render_async(cam, rps[i], backward=True) for i in range(2, pkg.FRAMES_IN_FLIGHT)]
    h1 = hip.render_async(cam, rps[0], backward=True)
    h2 = hip.render_async(cam, rps[1], backward=True)
    with pytest.raises(pkg.DrtHipError, match="frames are in flight"):
        hip.render_async(cam, rps[2], backward=True)
    with pytest.raises(pkg.DrtHipError, match="in flight"):
        hip.render(cam, rps[2], backward=True)
    with pytest.raises(pkg.DrtHipError, match="no such frame"):
        hip.wait((h2[0] + 5, h2[1], h2[2]))
    for hm in more:
        hip.wait(hm)
    a = hip.wait(h1)
    b = hip.wait(h2)
    np.testing.assert_array_equal(a[0], ref[0][0])
    np.testing.assert_array_equal(b[1], ref[1][1])
    # mesh scenes (the queue wavefront) and an adjoint image go through the same path
    mesh = pkg.scene_by_name("mesh10x12")
    hip.upload_scene(mesh)
    adj = np.random.RandomState(3).uniform(-1, 2, (64, 96, 3)).astype(np.float32)
    rref = hip.render(cam, rps[0], backward=True, adjoint=adj)
    h = hip.render_async(cam, rps[0], backward=True, adjoint=adj)
    r = hip.wait(h)
    np.testing.assert_array_equal(r[0], rref[0])
    np.testing.assert_array_equal(r[1], rref[1])
    # the copy launch's other shapes: a row-band shard (rows of other shards stay untouched: zero), a frame whose float
    # count is not a multiple of four, forward only, and frames of different sizes alternating between the two slots
    hip.upload_scene(scene)
    odd = pkg.cornell_camera(37, 29)
    for c, rp, bw in ((cam, pkg.RenderParams(spp=3, min_bounces=3, absorb=0.4, seed=8, shard=1, n_shards=3, band_rows=8), True),
                      (odd, pkg.RenderParams(spp=5, min_bounces=2, absorb=0.3, seed=9), True),
                      (odd, pkg.RenderParams(spp=5, min_bounces=2, absorb=0.3, seed=9, shard=2, n_shards=4, band_rows=4), True),
                      (cam, rps[2], False)):
        want = hip.render(c, rp, backward=bw)
        h_a = hip.render_async(c, rp, backward=bw)
        h_b = hip.render_async(cam, rps[3], backward=True)          # (the other slot, another size)
        got_a, got_b = hip.wait(h_a), hip.wait(h_b)
        np.testing.assert_array_equal(got_a[0], want[0])
        if bw:
            np.testing.assert_array_equal(got_a[1], want[1])
        assert got_a[2]["segments"] == want[2]["segments"]
        np.testing.assert_array_equal(got_b[0], ref[3][0])
        np.testing.assert_array_equal(got_b[1], ref[3][1])


def test_device_frames_overlap_and_stay_in_order(pkg):
    """Device-pointer renders that do not wait: the k_path grids of consecutive frames run on two streams of the context's
    own (alternating sets of partial-sum buffers), every frame's finishing launch on the context's stream in frame order.
    Results equal the synchronous call bit for bit -- separate output buffers, ONE output buffer written by every frame (the
    last frame must be what is left), a parameter update between two frames, a device adjoint image -- for the fixed-depth
    and the roulette-terminated kernel."""
    import torch
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)                 # (PyTorch's runtime before the library's, as in bench.py)
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(128, 96)
    r = pkg.HipRenderer(0)
    hip = pkg.HipRenderer(0)                   # (the synchronous renders it is compared with)
    try:
        r.upload_scene(scene)
        hip.upload_scene(scene)
        for b, p in ((4, 1.0), (1, 0.4)):
            rps = [pkg.RenderParams(spp=6, min_bounces=b, absorb=p, seed=s) for s in range(1, 8)]
            ref = [hip.render(cam, rp, backward=True) for rp in rps]
            outs = [torch.zeros((96, 128, 3), dtype=torch.float32, device=dev) for _ in rps]
            grads = [torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in rps]
            for rp, o, g in zip(rps, outs, grads):
                r.render_device(cam, rp, o.data_ptr(), g.data_ptr())
            r.synchronize()
            for (rimg, rgr, _), o, g in zip(ref, outs, grads):
                np.testing.assert_array_equal(o.cpu().numpy(), rimg)
                np.testing.assert_array_equal(g.cpu().numpy(), rgr)
            one_o, one_g = outs[0], grads[0]
            for rp in rps:                                     # every frame into the same buffers: frame order decides
                r.render_device(cam, rp, one_o.data_ptr(), one_g.data_ptr())
            r.synchronize()
            np.testing.assert_array_equal(one_o.cpu().numpy(), ref[-1][0])
            np.testing.assert_array_equal(one_g.cpu().numpy(), ref[-1][1])
        # a parameter update between two frames in flight, and a device adjoint image
        rp = pkg.RenderParams(spp=6, min_bounces=4, absorb=1.0, seed=3)
        newp = np.array(scene.params) * 0.6 + 0.1
        before = hip.render(cam, rp, backward=True)
        hip.update_params(newp)
        adj = np.random.RandomState(2).uniform(0.2, 1.4, (96, 128, 3)).astype(np.float32)
        after = hip.render(cam, rp, backward=True, adjoint=adj)
        o1, o2 = (torch.zeros((96, 128, 3), dtype=torch.float32, device=dev) for _ in range(2))
        g1, g2 = (torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in range(2))
        d_adj = torch.from_numpy(adj).to(dev)
        torch.cuda.synchronize()
        r.render_device(cam, rp, o1.data_ptr(), g1.data_ptr())
        r.update_params(newp)
        r.render_device(cam, rp, o2.data_ptr(), g2.data_ptr(), adjoint_ptr=d_adj.data_ptr())
        r.render_device(cam, rp, o1.data_ptr(), 0, backward=False)          # (forward only, the other set of buffers again)
        r.synchronize()
        np.testing.assert_array_equal(g1.cpu().numpy(), before[1])
        np.testing.assert_array_equal(o2.cpu().numpy(), after[0])
        np.testing.assert_array_equal(g2.cpu().numpy(), after[1])
        np.testing.assert_array_equal(o1.cpu().numpy(), after[0])
    finally:
        r.close()
        hip.close()


def test_every_user_of_the_partial_sum_buffers_is_ordered(pkg):
    """k_path's partial sums come in two sets ("lanes"); overlapped frames write theirs from streams of their own.  Sequences
    the C ABI allows and round 3 left unordered (the advisor's findings): (1) a gradient image rendered with device pointers
    and no wait -- lane 0 on the context's stream -- followed at once by frames that do not wait; (2) drt_hip_render_async
    straight after such frames; (3) DRT_RENDER_SERIAL frames between overlapping ones.  Frames large enough to still be
    running when the next call is made; every result must equal the synchronous call's bit for bit."""
    import ctypes as C
    import dataclasses
    import torch
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(384, 256)
    r = pkg.HipRenderer(0)
    hip = pkg.HipRenderer(0)
    try:
        r.upload_scene(scene)
        hip.upload_scene(scene)
        rps = [pkg.RenderParams(spp=24, min_bounces=6, absorb=1.0, seed=s) for s in range(11, 17)]
        ref = [hip.render(cam, rp, backward=True) for rp in rps]
        ref_gi = hip.render_gradient_image(cam, rps[0], 2)
        outs = [torch.zeros((256, 384, 3), dtype=torch.float32, device=dev) for _ in rps]
        grads = [torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in rps]
        gi_out = torch.zeros((256, 384, 3), dtype=torch.float32, device=dev)
        gi_img = torch.zeros((256, 384, 3), dtype=torch.float32, device=dev)
        for round_ in range(3):
            for t in outs + grads + [gi_out, gi_img]:
                t.zero_()
            torch.cuda.synchronize()
            # (1) frames in flight, then the gradient image (lane 0, context's stream, no wait), then frames again
            r.render_device(cam, rps[0], outs[0].data_ptr(), grads[0].data_ptr())
            r.render_device(cam, rps[1], outs[1].data_ptr(), grads[1].data_ptr())
            d = rps[0].to_desc()
            d.flags = pkg.RENDER_DEVICE_OUT
            cd = cam.to_desc()
            rc = r.lib.drt_hip_render_gradient_image(r.ctx, C.byref(cd), C.byref(d), 2, None, C.c_void_p(gi_out.data_ptr()),
                                                     C.c_void_p(gi_img.data_ptr()), None)
            assert rc == 0
            r.render_device(cam, rps[2], outs[2].data_ptr(), grads[2].data_ptr())
            # (3) a serial frame between overlapping ones
            r.render_device(cam, dataclasses.replace(rps[3], flags=rps[3].flags | pkg.RENDER_SERIAL), outs[3].data_ptr(), grads[3].data_ptr())
            r.render_device(cam, rps[4], outs[4].data_ptr(), grads[4].data_ptr())
            # (2) an asynchronous host-buffer frame straight after them
            h = r.render_async(cam, rps[5], backward=True)
            img5, g5, _ = r.wait(h)
            r.synchronize()
            for i in range(5):
                np.testing.assert_array_equal(outs[i].cpu().numpy(), ref[i][0], err_msg=f"frame {i}, round {round_}")
                np.testing.assert_array_equal(grads[i].cpu().numpy(), ref[i][1], err_msg=f"frame {i}, round {round_}")
            np.testing.assert_array_equal(img5, ref[5][0])
            np.testing.assert_array_equal(g5, ref[5][1])
            np.testing.assert_array_equal(gi_out.cpu().numpy(), ref_gi[0])
            np.testing.assert_array_equal(gi_img.cpu().numpy(), ref_gi[1])
    finally:
        r.close()
        hip.close()


def test_allreduce_on_the_second_stream_gives_the_same_gradients(pkg):
    """DRT_RENDER_ALLREDUCE_ASYNC (device buffers, a communicator): all-reduce and gradient copy run on the context's second
    stream, the steps alternate between two gradient sets; after drt_hip_synchronize every step's gradient equals the
    stream-ordered DRT_RENDER_ALLREDUCE result (a 1-rank communicator: the collective call is made like on 8 GPUs)."""
    import torch
    r = pkg.HipRenderer(0)
    try:
        scene = pkg.cornell_box()
        cam = pkg.cornell_camera(64, 48)
        r.upload_scene(scene)
        r.comm_init(pkg.comm_unique_id(), 0, 1)
        dev = torch.device("cuda", 0)
        out = torch.zeros((48, 64, 3), dtype=torch.float32, device=dev)
        seeds = [1, 2, 3, 4, 5]
        want = []
        for s in seeds:
            rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=s, flags=pkg.RENDER_ALLREDUCE)
            g = torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev)
            r.render_device(cam, rp, out.data_ptr(), g.data_ptr(), backward=True, sync=True)
            want.append(g.cpu().numpy())
        gs = [torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev) for _ in seeds]
        for s, g in zip(seeds, gs):                          # five steps enqueued back to back, no wait in between
            rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=s, flags=pkg.RENDER_ALLREDUCE_ASYNC)
            r.render_device(cam, rp, out.data_ptr(), g.data_ptr(), backward=True, sync=False)
        r.synchronize()
        torch.cuda.synchronize(dev)
        for g, w in zip(gs, want):
            np.testing.assert_array_equal(g.cpu().numpy(), w)
        # with DRT_RENDER_SYNC the call returns with the gradient in place
        g = torch.zeros((scene.n_params, 3), dtype=torch.float64, device=dev)
        rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=1, flags=pkg.RENDER_ALLREDUCE_ASYNC)
        st = r.render_device(cam, rp, out.data_ptr(), g.data_ptr(), backward=True, sync=True, want_stats=True)
        np.testing.assert_array_equal(g.cpu().numpy(), want[0])
        assert st["segments"] > 0
        # host buffers: the flag means DRT_RENDER_ALLREDUCE
        _, gh, _ = r.render(cam, rp, backward=True)
        np.testing.assert_array_equal(gh, want[0])
        r.comm_destroy()
    finally:
        r.close()


@pytest.mark.timeout(1200)
def test_bench_eight_ranks_plumbing():
    """The launch shape of the driver's SCALE run -- 8 ranks, interleaved 16-row bands, weak scaling -- on one device over a
    gloo rendezvous (RCCL refuses several ranks on one GPU: the reduce falls back to torch.distributed here, the only
    place that fallback exists); the numbers of such a run mean nothing."""
    line = _bench("--gpus", "8", "--dist-backend", "gloo", "--same-gpu", "--steps", "2", "--warmup", "1",
                  "--width", "128", "--height", "128", "--spp", "2", "--no-cpu-baseline", "--no-extra-views", timeout=1100)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert "8 ranks" in line["config"]["parallelism"]
    assert line["config"]["paths_per_step"] == 8 * 128 * 128 * 2
    assert "weak_scaling" in line


def test_pinned_caller_buffers_get_the_same_bits(pkg):
    """drt_hip_pin_host (ABI v7): a render whose out_rgb (or gradient image) lies inside a pinned range is written there by the
    finishing kernel -- same bits as through the staging block, for k_path, the queue wavefront, shards, the gradient image
    and asynchronous frames; overlapping ranges and unknown pointers are refused; after unpin the buffer is an ordinary one."""
    r = pkg.HipRenderer(0)
    try:
        scene = pkg.cornell_box()
        cam = pkg.cornell_camera(96, 64)
        r.upload_scene(scene)
        rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=5, band_rows=8)
        adj = np.random.RandomState(2).uniform(0, 1, (64, 96, 3)).astype(np.float32)
        ref, gref, _ = r.render(cam, rp, backward=True, adjoint=adj)
        block = np.zeros((3, 64, 96, 3), dtype=np.float32)        # one pinned range, three images in it
        r.pin_host(block)
        with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
            r.pin_host(block[1])                                   # overlaps
        img, g, st = r.render(cam, rp, backward=True, adjoint=adj, img_out=block[1])
        assert np.shares_memory(img, block[1]) and st["segments"] > 0
        np.testing.assert_array_equal(img, ref)
        np.testing.assert_array_equal(g, gref)
        assert not block[0].any() and not block[2].any()
        # the adjoint itself may be pinned too (no staging copy)
        block[2][:] = adj
        img, g, _ = r.render(cam, rp, backward=True, adjoint=block[2], img_out=block[0], want_stats=False)
        np.testing.assert_array_equal(img, ref)
        np.testing.assert_array_equal(g, gref)
        # the queue wavefront and a shard of the frame
        rq = dataclasses.replace(rp, bounces_per_launch=1, shard=1, n_shards=2)
        want, gq, _ = r.render(cam, rq, backward=True)
        block[0][:] = -1.0
        img, g, _ = r.render(cam, rq, backward=True, img_out=block[0])
        rows = pkg.shard_rows(64, 8, 2, 1)
        np.testing.assert_array_equal(img[rows], want[rows])
        np.testing.assert_array_equal(g, gq)
        other = np.setdiff1d(np.arange(64), rows)
        assert (img[other] == -1.0).all()                          # rows of the other shard are not touched
        # asynchronous frames into pinned buffers
        hs = [r.render_async(cam, dataclasses.replace(rp, seed=s), backward=True, img_out=block[i]) for i, s in enumerate((7, 8, 9))]
        outs = [r.wait(h) for h in hs]
        for (im, gg, _), s in zip(outs, (7, 8, 9)):
            w, gw, _ = r.render(cam, dataclasses.replace(rp, seed=s), backward=True)
            np.testing.assert_array_equal(im, w)
            np.testing.assert_array_equal(gg, gw)
        with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
            r.unpin_host(block[1])                                 # not the start of a pinned range
        r.unpin_host(block)
        img, g, _ = r.render(cam, rp, backward=True, adjoint=adj, img_out=block[1])
        np.testing.assert_array_equal(img, ref)
    finally:
        r.close()


def test_update_params_is_ordered_before_overlapped_frames(pkg):
    """drt_hip_update_params installs the new values with a launch in the context's stream and returns; frames that do not wait
    (device pointers, asynchronous host frames) run their path kernels on streams of their own.  They must see the NEW values:
    a scene with an albedo per face -- 19,804 parameters, an install launch of some tens of microseconds -- rendered right
    behind every update, forward only and without an adjoint (nothing else orders the frame behind the context's stream),
    against the synchronous call bit for bit.  And an asynchronous frame keeps its own copy of a PINNED adjoint image."""
    import torch
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    scene = pkg.scene_by_name("mesh100x100fall")
    assert scene.n_params == 4 + 19800
    cam = pkg.cornell_camera(96, 64)
    rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=5)
    rs = np.random.RandomState(11)
    base = np.asarray(scene.params, dtype=np.float64)
    sets = []
    for k in range(2):
        p = base.copy()
        p[4:] = rs.uniform(0.05 if k else 0.6, 0.35 if k else 0.95, (len(base) - 4, 3))      # (dark faces / bright faces: any stale value shows)
        sets.append(p)
    r = pkg.HipRenderer(0)
    try:
        r.upload_scene(scene)
        want = []
        for p in sets:
            r.update_params(p)
            img, _, st = r.render(cam, rp, backward=False)
            assert st["kernels"]["path"]["launches"] == 1          # (k_path_mesh: the route whose frames overlap)
            want.append(img.copy())
        assert np.abs(want[0] - want[1]).max() > 1e-2
        out = [torch.zeros((64, 96, 3), dtype=torch.float32, device=dev) for _ in range(2)]
        for it in range(24):
            k = it & 1
            r.update_params(sets[k])
            r.render_device(cam, rp, out[k].data_ptr(), 0, backward=False)             # returns after enqueueing
            if it % 5 == 4:
                r.update_params(sets[k ^ 1])                                            # (two updates in a row, two frames behind them)
                r.render_device(cam, rp, out[k ^ 1].data_ptr(), 0, backward=False)
                r.render_device(cam, rp, out[k ^ 1].data_ptr(), 0, backward=False)
                r.synchronize()
                torch.cuda.synchronize(dev)
                np.testing.assert_array_equal(out[k ^ 1].cpu().numpy(), want[k ^ 1])
            r.synchronize()
            torch.cuda.synchronize(dev)
            np.testing.assert_array_equal(out[k].cpu().numpy(), want[k])
        # asynchronous host frames: the same, and a pinned adjoint image may be rewritten as soon as render_async has returned
        scene2 = pkg.cornell_box()
        r.upload_scene(scene2)
        cam2 = pkg.cornell_camera(64, 48)
        rp2 = pkg.RenderParams(spp=4, min_bounces=3, absorb=1.0, seed=2)
        adj = np.ascontiguousarray(rs.uniform(-1, 2, (48, 64, 3)).astype(np.float32))
        r.pin_host(adj)
        a0 = adj.copy()
        _, g_want, _ = r.render(cam2, rp2, backward=True, adjoint=a0)
        handles = []
        for i in range(3):
            adj[...] = a0
            handles.append(r.render_async(cam2, rp2, backward=True, adjoint=adj))
            adj[...] = 1e6                                                              # the caller's next frame overwrites it
        for h in handles:
            _, g, _ = r.wait(h)
            np.testing.assert_array_equal(g, g_want)
        r.unpin_host(adj)
    finally:
        r.close()


def test_group_over_all_visible_devices_equals_one_context(pkg, hip):
    """First contact with a multi-GPU box: drt_hip_create_group over ALL visible devices (distinct devices: ONE grouped
    ncclAllReduce over xGMI inside the library) against the single-context render -- image bit for bit (disjoint 16-row bands),
    gradients to fp64 summation order.  Skipped where one device is visible (every box so far)."""
    n = pkg.load_library().drt_hip_device_count()
    if n < 2:
        pytest.skip("one visible device")
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(256, 192)
    rp = pkg.RenderParams(spp=8, min_bounces=6, absorb=1.0, seed=4, band_rows=16)
    hip.upload_scene(scene)
    want_img, want_g, want_st = hip.render(cam, rp, backward=True)
    g = pkg.HipRenderer(list(range(n)))
    try:
        assert g.group_size == n and len({g.pci_bus_id(i) for i in range(n)}) == n
        g.upload_scene(scene)
        for _ in range(3):
            img, grads, st = g.render(cam, rp, backward=True)
            np.testing.assert_array_equal(img, want_img)
            assert st["segments"] == want_st["segments"]
            assert np.abs(grads - want_g).max() <= 1e-6 * np.abs(want_g).max()
        # ... and a per-face scene (a 1.2 MB gradient vector through the same all-reduce) on the first two devices
        mesh = pkg.scene_by_name("mesh40x40fall")
        hip.upload_scene(mesh)
        cam2 = pkg.cornell_camera(128, 96)
        rp2 = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=6, band_rows=16)
        _, mg, _ = hip.render(cam2, rp2, backward=True)
        g.upload_scene(mesh)
        _, gg, _ = g.render(cam2, rp2, backward=True)
        assert np.abs(gg - mg).max() <= 1e-6 * np.abs(mg).max()
    finally:
        g.close()
