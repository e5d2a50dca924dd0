"""Triangle meshes on the device (extension behind the Shape plugin surface; BASELINE config 4
shape): the BVH traversal of K2 against the brute-force oracle, whose triangle semantics are
pinned by a Triangle plugin running inside the unmodified reference (tests/golden/g9..g11)."""
import dataclasses
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def grad_rel_err(got, want):
    return float(np.abs(got - want).max() / np.abs(want).max())


def test_50k_triangle_bvh_matches_brute_force(pkg, hip, oracle):
    """160 x 160 displaced sphere = 50,880 triangles: every closest hit found through the BVH must
    be the one the oracle's linear scan finds (f64 mode: identical paths, gradients to 1e-9)."""
    scene = pkg.cornell_with_mesh(160, 160)
    assert len(scene.meshes[0][1]) == 50880
    cam = pkg.cornell_camera(40, 32)
    rp = pkg.RenderParams(spp=2, min_bounces=4, absorb=1.0, seed=6)
    ref = oracle.render(scene, cam, rp, backward=True)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True, f64=True)
    assert st["segments"] == ref["stats"]["segments"]
    assert grad_rel_err(grads, ref["grads"]) < 1e-9
    np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    img, grads, st = hip.render(cam, rp, backward=True)
    assert abs(st["segments"] - ref["stats"]["segments"]) <= 64
    scale = np.abs(ref["image"]).max()
    bad = np.abs(img.astype(np.float64) - ref["image"]).max(-1) > 2e-4 * scale
    assert bad.sum() <= 1          # (one pixel may hold an f32-flipped path: tests/test_gpu_parity.py)
    assert grad_rel_err(grads, ref["grads"]) <= 1e-4


@pytest.mark.parametrize("name,W,H,spp,kw", [
    ("mesh10x12", 24, 24, 3, dict(min_bounces=4, absorb=1.0)),
    ("mesh40x40", 48, 48, 4, dict(min_bounces=5, absorb=1.0)),
    ("mesh10x12", 29, 28, 3, dict(min_bounces=1, absorb=0.5)),                     # the reference's default roulette
    ("mesh40x40", 40, 30, 4, dict(min_bounces=2, absorb=0.3, max_depth=6)),        # a user cap that cuts paths short
    ("mesh10x12", 33, 17, 5, dict(min_bounces=0, absorb=0.25, max_depth=12)),      # roulette at depth 0
])
def test_one_launch_mesh_route_against_the_oracle_and_the_queue_route(pkg, hip, oracle, name, W, H, spp, kw):
    """Small frames of a mesh scene take k_path_mesh (csrc/drt_path_mesh.h): the whole path in one launch, the BVH walk inside.
    f64: the oracle's segment and capped-path counts exactly, gradients to 1e-9, image to f32 rounding; f32: the f32 bounds;
    and the queue wavefront (bounces_per_launch = 1) walks the same candidates."""
    scene = pkg.scene_by_name(name)
    cam = pkg.cornell_camera(W, H)
    rp = pkg.RenderParams(spp=spp, seed=3, **kw)
    rq = dataclasses.replace(rp, bounces_per_launch=1)
    hip.upload_scene(scene)
    ref = oracle.render(scene, cam, rp, backward=True)
    for f64 in (True, False):
        img, grads, st = hip.render(cam, rp, backward=True, f64=f64)
        imq, gq, stq = hip.render(cam, rq, backward=True, f64=f64)
        assert st["kernels"]["path"]["launches"] == 1 and st["kernels"]["shade"]["launches"] == 0
        assert stq["kernels"]["path"]["launches"] == 0 and stq["kernels"]["intersect_mesh"]["launches"] > 0
        if f64:
            assert st["segments"] == ref["stats"]["segments"] == stq["segments"]
            assert st["capped_paths"] == stq["capped_paths"]
            assert st["kernels"]["intersect_mesh"]["units"] == stq["kernels"]["intersect_mesh"]["units"]   # rays that reach the mesh bounds
            assert grad_rel_err(grads, ref["grads"]) < 1e-9
            np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
        else:
            assert abs(st["segments"] - ref["stats"]["segments"]) <= 8
            assert grad_rel_err(grads, ref["grads"]) <= 1e-4 and grad_rel_err(grads, gq) <= 1e-4
            bad = np.abs(img.astype(np.float64) - ref["image"]).max(-1) > 2e-4 * np.abs(ref["image"]).max()
            assert bad.sum() <= 1


def test_one_launch_mesh_route_adjoint_gradient_image_and_shards(pkg, hip):
    """k_path_mesh with a per-pixel adjoint, as a gradient image, and as shards of the frame: the queue route's numbers to f32
    rounding, the shards tile the frame bit for bit and their gradients add up."""
    scene = pkg.scene_by_name("mesh40x40")
    cam = pkg.cornell_camera(64, 48)
    rp = pkg.RenderParams(spp=6, min_bounces=4, absorb=1.0, seed=9, band_rows=4)
    rq = dataclasses.replace(rp, bounces_per_launch=1)
    hip.upload_scene(scene)
    rs = np.random.RandomState(5)
    adj = rs.uniform(0.0, 2.0, (48, 64, 3)).astype(np.float32)
    img, g, st = hip.render(cam, rp, backward=True, adjoint=adj)
    imq, gq, _ = hip.render(cam, rq, backward=True, adjoint=adj)
    assert st["kernels"]["path"]["launches"] == 1
    assert grad_rel_err(g, gq) < 2e-5
    np.testing.assert_allclose(img, imq, rtol=2e-5, atol=1e-7)
    white = scene.param_names.index("white")
    _, gi, sti = hip.render_gradient_image(cam, rp, white)
    _, giq, _ = hip.render_gradient_image(cam, rq, white)
    assert sti["kernels"]["path"]["launches"] == 1
    np.testing.assert_allclose(gi, giq, rtol=1e-4, atol=1e-6 * float(np.abs(giq).max()))
    img1, g1, _ = hip.render(cam, rp, backward=True)
    acc, gsum = np.zeros_like(img1), np.zeros_like(g1)
    for s in range(3):
        im_s, g_s, st_s = hip.render(cam, dataclasses.replace(rp, shard=s, n_shards=3), backward=True)
        assert st_s["kernels"]["path"]["launches"] == 1
        rows = pkg.shard_rows(48, rp.band_rows, 3, s)
        acc[rows] = im_s[rows]
        gsum += g_s
    np.testing.assert_array_equal(acc, img1)
    np.testing.assert_allclose(gsum, g1, rtol=1e-9)
    # forward only, and twice the same bits
    f1, _, stf = hip.render(cam, rp, backward=False)
    f2, _, _ = hip.render(cam, rp, backward=False)
    assert stf["kernels"]["path"]["launches"] == 1
    np.testing.assert_array_equal(f1, f2)
    np.testing.assert_array_equal(f1, img1)


def test_per_face_parameters_use_the_general_gradient_path(pkg, hip, oracle):
    """12 per-face albedos + the 4 Cornell parameters = 16 > 8: K6's LDS columns + fp64 atomics."""
    scene = pkg.cornell_with_mesh(24, 24, per_face_params=12)
    assert scene.n_params == 16
    cam = pkg.cornell_camera(48, 48)
    rp = pkg.RenderParams(spp=4, min_bounces=2, absorb=0.3, seed=8)
    ref = oracle.render(scene, cam, rp, backward=True)
    hip.upload_scene(scene)
    _, g64, st = hip.render(cam, rp, backward=True, f64=True)
    assert st["segments"] == ref["stats"]["segments"]
    np.testing.assert_allclose(g64, ref["grads"], rtol=1e-9, atol=1e-12)
    _, g32, _ = hip.render(cam, rp, backward=True)
    assert grad_rel_err(g32, ref["grads"]) <= 1e-4
    assert np.abs(ref["grads"][4:]).min() > 0        # every face parameter is reached


def test_an_albedo_parameter_per_face(pkg, hip, oracle):
    """drt_mesh_desc::face_param: every face of the mesh a BxDF of its own (config 4 as SURVEY 8d words it) -- 1104 + 4
    parameters here: beyond the 256 that K6 keeps in LDS, so the fp64 atomics on the gradient vector run too.  The oracle's
    face_param is pinned, bit for bit, to a material per face (tests/test_oracle_golden.py), which the reference harness runs."""
    scene = pkg.scene_by_name("mesh24x24fall")
    assert scene.n_params == 4 + 2 * 24 * 23
    cam = pkg.cornell_camera(56, 48)
    rp = pkg.RenderParams(spp=4, min_bounces=3, absorb=0.2, seed=4, max_depth=7)
    ref = oracle.render(scene, cam, rp, backward=True)
    hip.upload_scene(scene)
    i64, g64, st = hip.render(cam, rp, backward=True, f64=True)
    assert st["kernels"]["path"]["launches"] == 0            # more than 8 parameters: the tape route
    assert st["segments"] == ref["stats"]["segments"]
    np.testing.assert_allclose(g64, ref["grads"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(i64, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    _, g32, _ = hip.render(cam, rp, backward=True)
    assert grad_rel_err(g32, ref["grads"]) <= 1e-4
    reached = np.abs(ref["grads"][4:]).sum(1) > 0
    assert reached.sum() > 100                                # a couple of hundred faces see light in this frame
    # faces no path met keep an exactly zero gradient on the device too
    assert not g64[4:][~reached].any() and not g32[4:][~reached].any()
    # a parameter that does not require grad stays zero, its neighbours are untouched
    scene.requires_grad[4 + int(np.argmax(reached))] = False
    hip.upload_scene(scene)
    _, g2, _ = hip.render(cam, rp, backward=True, f64=True)
    k = 4 + int(np.argmax(reached))
    assert not g2[k].any()
    np.testing.assert_allclose(np.delete(g2, k, 0), np.delete(g64, k, 0), rtol=1e-12)      # (fp64 atomics: the order of the adds varies)
    # the gradient image of one face's albedo
    _, gi, _ = hip.render_gradient_image(cam, rp, 4 + int(np.argmax(np.abs(ref["grads"][4:]).sum(1))), f64=True)
    assert np.isfinite(gi).all() and np.abs(gi).max() > 0


def test_mesh_scene_properties_at_scale(pkg, hip):
    """Config-4-shaped render (50k triangles, 256 x 256 x 16): linearity in emission, determinism,
    sharding."""
    scene = pkg.cornell_with_mesh(160, 160)
    cam = pkg.cornell_camera(256, 256)
    rp = pkg.RenderParams(spp=16, min_bounces=6, absorb=1.0, seed=2)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True)
    total = img.astype(np.float64).sum((0, 1)) * rp.spp
    np.testing.assert_allclose(grads[3] * np.array(scene.params[3]), total, rtol=2e-6)
    img2, grads2, st2 = hip.render(cam, rp, backward=True)
    np.testing.assert_array_equal(img, img2)
    np.testing.assert_array_equal(grads, grads2)
    acc = np.zeros_like(img)
    gsum = np.zeros_like(grads)
    for s in range(2):
        im_s, g_s, _ = hip.render(cam, dataclasses.replace(rp, shard=s, n_shards=2), backward=True)
        rows = pkg.shard_rows(256, rp.band_rows, 2, s)
        acc[rows] = im_s[rows]
        gsum += g_s
    np.testing.assert_array_equal(acc, img)
    np.testing.assert_allclose(gsum, grads, rtol=1e-9)


def test_mesh_errors(pkg, hip):
    scene = pkg.cornell_with_mesh(6, 8)
    v, idx, fm = scene.meshes[0]
    bad = idx.copy()
    bad[3, 1] = len(v) + 5
    scene.meshes[0] = (v, bad, fm)
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        hip.upload_scene(scene)
    scene.meshes[0] = (v, idx, np.full(len(idx), 99, np.int32))
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        hip.upload_scene(scene)
    hip.upload_scene(pkg.cornell_box())          # the context is still usable


@pytest.mark.gpu
def test_two_stage_shade_launches_change_no_bit():
    """DRT_HIP_TAIL_BOUNCES=2 (a shade launch keeps the rays that miss the bounds of the mesh in registers through one more vertex;
    the unbiased operator's rounds run that way) against =1: same image, gradients, segment / capped / walked counts on ten
    renders x f64 and f32 x whole frame, shard, small batches; f64 against the oracle.  The knob is read once per process:
    tools/tail_check.py renders each setting in a process of its own."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "tail_check.py"), "parity"], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "parity ok" in out.stdout and out.stdout.count("\nok ") + out.stdout.startswith("ok ") == 60
