"""Parity of the HIP wavefront path (through the C ABI) against the golden vectors captured from
the unmodified reference and against the fp64 oracle.  All tests need a real MI355X.

Stated tolerances (north star: fp32 forward tolerance, gradients within 1e-4 relative):
  * f64 device mode vs oracle/reference: 1e-9 relative on image and gradients (same algorithm,
    same RNG draws; only FMA contraction / libm differ).
  * f32 device mode, the reference's scenes: gradients 1e-4 of the largest gradient component, image
    mean 1e-4 relative, every pixel within 2e-4 * max(image) -- except pixels that contain a path whose
    DISCRETE decision (which surface is hit) flipped under f32 rounding: at most one such pixel per 100,000
    paths, never more than FLIP_BUDGET_MIN = 1 on the small fixtures; segment count within one path's
    length.  Measured at HEAD (profiles/r02_parity_report.txt): gradients <= 4e-6, pixels <= 3e-5, and
    one flipped path in one fixture (g13, a gradient image at 6 spp).
  * f32, the RANDOM test scenes only (g7, g8, random<seed>: roulette-boosted long paths, exponent-80
    lobes, non-unit wall normals -- per-path weights span six orders of magnitude):
    a single path whose discrete hit decision flips under f32 rounding moves a small frame's gradient by
    up to ~1e-3 (measured: 8.3e-4 for one silhouette-edge path among 15,360) and a channel's mean by 14 %.
    The fixtures g7 / g8 (no flip in their sample sets: 9.5e-7 / 3.1e-7 measured) keep a gradient bound of
    2e-4 and an outlier budget of 0.5 % of the pixels; scenes rendered live go through check_f32_heavy_tailed,
    which sets the flipped pixels aside (same budget) and removes their share from mean and gradient before
    it compares.  The f64 mode, flip-free, is what pins those scenes.
"""
import os

import numpy as np
import pytest

from conftest import LOSS_GOLDENS, SMALL_GOLDENS, case_inputs, load_golden

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4
GRAD_TOL_HEAVY = 2e-4        # random scenes only: 2 x the 8.3e-5 measured on g7 (see above)
MEAN_TOL = 1e-4
PIXEL_TOL = 2e-4
OUTLIER_FRAC_HEAVY = 5e-3    # random scenes only
FLIP_MIN_REL = 1e-2          # a set-aside pixel must be off by at least this fraction of its own value (a flipped path, not rounding)


def flip_budget(n_paths):
    """Pixels that may contain an f32-flipped path: one per 100,000 paths, at least one."""
    return max(1, int(n_paths // 100000))


def grad_rel_err(got, want):
    return float(np.abs(got - want).max() / np.abs(want).max())


def check_f32(img, grads, segments, g_img, g_grads, g_segments, heavy_tailed=False, n_paths=0):
    """The f32 device mode against reference numbers.  No self-widening terms: the bounds are the stated ones."""
    scale = float(np.abs(g_img).max())
    bad = np.abs(img.astype(np.float64) - g_img).max(-1) > PIXEL_TOL * scale
    if heavy_tailed:
        assert bad.mean() <= OUTLIER_FRAC_HEAVY, f"{bad.sum()} of {bad.size} pixels outside fp32 tolerance"
    else:
        assert bad.sum() <= flip_budget(n_paths), f"{bad.sum()} of {bad.size} pixels outside fp32 tolerance"
    m_got, m_want = img.astype(np.float64).mean((0, 1)), g_img.mean((0, 1))
    assert np.abs(m_got - m_want).max() <= (5 * MEAN_TOL if heavy_tailed else MEAN_TOL) * m_want.max()
    # one path whose fp32 hit/miss decision flips can change the count by its whole length (<= 64)
    assert abs(int(segments) - int(g_segments)) <= (max(64, int(2e-4 * g_segments)) if heavy_tailed else 64)
    if g_grads is not None:
        assert grad_rel_err(grads, g_grads) <= (GRAD_TOL_HEAVY if heavy_tailed else GRAD_TOL)


def check_f32_heavy_tailed(hip, cam, rp, adjoint, n_params, ref_img, ref_grads, ref_segments, ref_grad_image):
    """The f32 mode on the RANDOM scenes, where a single path can weigh 1e5 times the average (exponent-80 lobes, roulette
    boosts): a path whose discrete hit decision flips under f32 rounding then moves the frame's mean or a gradient by far
    more than any rounding bound (measured: one silhouette-edge path of a 15,360-path frame, 8.3e-4 of the largest gradient
    component; one of a 61,440-path frame, 14 % of a channel's mean).  So the check is flip-aware: pixels outside the f32
    pixel bound are counted (at most 0.5 % of the frame) and SET ASIDE -- the mean is taken over the others, and the
    gradient is compared after removing what the set-aside pixels contribute: in f32 as the device's per-pixel gradient
    image of each parameter gives it, on the reference side as the CHECKER's per-pixel gradient image gives it
    (ref_grad_image(p): nothing the device produced excuses the device)."""
    img, grads, st = hip.render(cam, rp, backward=True, adjoint=adjoint)
    scale = float(np.abs(ref_img).max())
    bad = np.abs(img.astype(np.float64) - ref_img).max(-1) > PIXEL_TOL * scale
    assert bad.mean() <= OUTLIER_FRAC_HEAVY, f"{bad.sum()} of {bad.size} pixels outside fp32 tolerance"
    assert abs(int(st["segments"]) - int(ref_segments)) <= max(64, int(2e-4 * ref_segments))
    # ... and only a DISCRETE difference sets a pixel aside: a path whose decision flipped is there or not, a whole sample's
    # weight of the pixel's few -- at least a per cent of the pixel's own value (measured on these scenes: 4 % and up) -- while a
    # rounding defect is parts in 1e6..1e4 of it.  A pixel off by more than the bound but by less than that is NOT excused.
    if bad.any():
        d = np.abs(img.astype(np.float64) - ref_img)[bad].max(-1)
        own = np.maximum(np.abs(ref_img)[bad].max(-1), np.abs(img.astype(np.float64))[bad].max(-1))
        assert (d >= FLIP_MIN_REL * own).all(), ("a set-aside pixel differs by a rounding-sized amount", (d / own).min())
    good = ~bad
    m_got, m_want = img.astype(np.float64)[good].mean(0), ref_img[good].mean(0)
    assert np.abs(m_got - m_want).max() <= 5 * MEAN_TOL * m_want.max()
    corr = np.zeros_like(ref_grads)
    if bad.any():
        for p in range(n_params):
            _, gi32, _ = hip.render_gradient_image(cam, rp, p, adjoint=adjoint)
            corr[p] = (gi32.astype(np.float64) - ref_grad_image(p))[bad].sum(0) * rp.spp
    assert grad_rel_err(grads - corr, ref_grads) <= GRAD_TOL_HEAVY, (int(bad.sum()), grad_rel_err(grads, ref_grads))
    return int(bad.sum())


@pytest.mark.parametrize("name", SMALL_GOLDENS)
def test_f32_matches_reference_golden(pkg, hip, name):
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    img, grads, stats = hip.render(cam, rp, backward=True, adjoint=adjoint)
    check_f32(img, grads, stats["segments"], g["image"], g["grads"], g["segments"], heavy_tailed="random" in name)


@pytest.mark.parametrize("name", SMALL_GOLDENS)
def test_f64_mode_matches_reference_golden(pkg, hip, name):
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    img, grads, stats = hip.render(cam, rp, backward=True, adjoint=adjoint, f64=True)
    assert stats["segments"] == int(g["segments"])
    assert grad_rel_err(grads, g["grads"]) < 1e-9
    # the image comes back as float32: compare at float32 resolution
    np.testing.assert_allclose(img, g["image"].astype(np.float32), rtol=2e-7, atol=1e-12)


@pytest.mark.parametrize("name", LOSS_GOLDENS)
def test_per_sample_squared_error_loss_matches_the_references_autograd(pkg, hip, name):
    """DRT_RENDER_LOSS_L2: every camera sample back-propagated through a loss of its own, |L_s - target_pixel|^2 -- the loop of
    the reference's README.md:93-98 with loss_func = squared error.  Fixtures from the reference's own autograd
    (`diff = radiance - target; loss = diff * diff; loss.backward(1)`): f64 mode to 1e-9 with identical segment counts, f32 to
    the stated gradient bound; group contexts and batches change nothing; and the flag is refused where it means nothing."""
    import dataclasses
    g = load_golden(name)
    scene, cam, rp, target = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True, adjoint=target, f64=True, loss_l2=True)
    assert st["segments"] == int(g["segments"])
    assert grad_rel_err(grads, g["grads"]) < 1e-9
    np.testing.assert_allclose(img, g["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    # under the default (DRT_SPECIALISE_AUTO) no frame waits for a compiler: the tape route renders until the LOSS kernel has
    # arrived from the library's compile thread -- either way within the bound
    _, g_auto, _ = hip.render(cam, rp, backward=True, adjoint=target, loss_l2=True)
    assert grad_rel_err(g_auto, g["grads"]) <= (GRAD_TOL_HEAVY if "random" in name else GRAD_TOL)
    hip.set_specialisation(pkg.SPECIALISE_NOW)
    try:
        img32, g32, st32 = hip.render(cam, rp, backward=True, adjoint=target, loss_l2=True)
    finally:
        hip.set_specialisation(pkg.SPECIALISE_AUTO)
    assert grad_rel_err(g32, g["grads"]) <= (GRAD_TOL_HEAVY if "random" in name else GRAD_TOL)
    # f32, analytic scene, no shape with both a BxDF and an emitter: ONE k_path launch, its LOSS instantiation compiled at run
    # time (the path's radiance is final where it meets the light); otherwise the tape route -- and the two agree
    on_path = not scene.meshes and not any(m >= 0 and e >= 0 for _, m, e, _ in scene.shapes)
    assert (st32["kernels"]["path"]["launches"] == 1) == on_path, st32["kernels"]
    assert st32["path_program"] == ("specialised" if on_path else "none")
    _, gq, stq = hip.render(cam, dataclasses.replace(rp, bounces_per_launch=1), backward=True, adjoint=target, loss_l2=True)
    assert stq["kernels"]["path"]["launches"] == 0
    np.testing.assert_allclose(g32, gq, rtol=1e-4 if "random" in name else 2e-5, atol=1e-5 * np.abs(gq).max())
    # several batches, several bounces per launch: the same paths, sums equal to rounding
    _, gb, _ = hip.render(cam, dataclasses.replace(rp, batch_paths=1000, bounces_per_launch=2), backward=True, adjoint=target,
                          f64=True, loss_l2=True)
    assert grad_rel_err(gb, g["grads"]) < 1e-9
    with pytest.raises(pkg.DrtHipError, match="DRT_RENDER_LOSS_L2"):
        hip.render(cam, rp, backward=True, loss_l2=True)                              # no target image
    with pytest.raises(pkg.DrtHipError, match="DRT_RENDER_LOSS_L2"):
        hip.render(cam, rp, backward=True, adjoint=target, loss_l2=True, unbiased=True)


def test_config1_256x256x8_depth4(pkg, hip):
    """BASELINE config 1 (the reference's own CPU-runnable case), full image from the reference."""
    g = load_golden("c1_cornell_256x256x8_d4")
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    img, grads, stats = hip.render(cam, rp, backward=True)
    check_f32(img, grads, stats["segments"], g["image"].astype(np.float64), g["grads"], g["segments"])
    img64, grads64, stats64 = hip.render(cam, rp, backward=True, f64=True)
    assert stats64["segments"] == int(g["segments"])
    assert grad_rel_err(grads64, g["grads"]) < 1e-9


def test_config3_512x512x64_depth8_full_size(pkg, hip, oracle):
    """BASELINE config 2/3 at full size against numbers produced by the reference itself
    (116 s of its CPU time): parameter gradients within 1e-4, mean radiance, 8x8 block means -- and four whole rows of the
    headline's own frame, at its 64 spp, PER PIXEL against the checker run live (through the light, along the ceiling where
    its light falls, through the back sphere, through the front sphere): exact in the f64 mode, PIXEL_TOL with at most one
    flipped pixel per row in f32; each of those rows rendered alone is that row of the full frame bit for bit."""
    g = load_golden("c3_cornell_512x512x64_d8")
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    img, grads, stats = hip.render(cam, rp, backward=True)
    assert stats["paths"] == 512 * 512 * 64
    assert abs(stats["segments"] - int(g["segments"])) <= 2e-5 * int(g["segments"])
    assert grad_rel_err(grads, g["grads"]) <= GRAD_TOL
    im = img.astype(np.float64)
    np.testing.assert_allclose(im.mean((0, 1)), g["mean_rgb"], rtol=MEAN_TOL)
    blocks = im.reshape(64, 8, 64, 8, 3).mean((1, 3))
    assert np.abs(blocks - g["image_block_mean"]).max() <= 2e-3 * g["image_block_mean"].max()
    np.testing.assert_allclose(im.mean(1), g["row_mean"], rtol=0, atol=2e-4 * g["row_mean"].max())
    # forward-only call gives the same image bit for bit and the same segment count
    img_f, _, stats_f = hip.render(cam, rp, backward=False)
    assert stats_f["segments"] == stats["segments"]
    np.testing.assert_array_equal(img_f, img)
    import dataclasses
    for row in (40, 100, 188, 256):
        rr = dataclasses.replace(rp, shard=row, n_shards=512, band_rows=1)
        ref = oracle.render(scene, cam, rr, backward=True)
        i64, g64, s64 = hip.render(cam, rr, backward=True, f64=True)
        assert s64["segments"] == ref["stats"]["segments"]
        assert grad_rel_err(g64, ref["grads"]) < 1e-9
        np.testing.assert_allclose(i64[row], ref["image"][row].astype(np.float32), rtol=2e-7, atol=1e-12)
        i32, g32, s32 = hip.render(cam, rr, backward=True)
        np.testing.assert_array_equal(i32[row], img[row])
        assert abs(s32["segments"] - ref["stats"]["segments"]) <= 64
        assert grad_rel_err(g32, ref["grads"]) <= GRAD_TOL
        off = np.abs(img[row].astype(np.float64) - ref["image"][row]).max(-1) > PIXEL_TOL * np.abs(ref["image"][row]).max()
        assert off.sum() <= 1, (row, int(off.sum()))
    assert img[40].max() > 0.9 and img[256, 256].max() < img[40].max()      # (row 40 does cross the light)


def test_linearity_in_emission_full_size(pkg, hip):
    """Radiance is linear in the emission parameter: sum(grad_E * E) == sum of all path radiances
    (SURVEY 4.3), checked at 512x512x16."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(512, 512)
    rp = pkg.RenderParams(spp=16, min_bounces=6, absorb=1.0, seed=5)
    hip.upload_scene(scene)
    img, grads, _ = hip.render(cam, rp, backward=True)
    total = img.astype(np.float64).sum((0, 1)) * rp.spp
    e = scene.param_names.index("emission")
    np.testing.assert_allclose(grads[e] * np.array(scene.params[e]), total, rtol=2e-6)


def test_deterministic_and_batch_independent(pkg, hip):
    scene = pkg.cornell_box(front_specular=True)
    cam = pkg.cornell_camera(96, 64)
    rp = pkg.RenderParams(spp=12, min_bounces=2, absorb=0.4, seed=9)
    hip.upload_scene(scene)
    a = hip.render(cam, rp, backward=True)
    b = hip.render(cam, rp, backward=True)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])          # fixed-order reductions: bitwise equal
    assert a[2]["segments"] == b[2]["segments"]
    import dataclasses
    for batch in (1000, 96 * 64, 96 * 64 * 5 + 7):
        c = hip.render(cam, dataclasses.replace(rp, batch_paths=batch), backward=True)
        assert c[2]["segments"] == a[2]["segments"] and c[2]["batches"] > 1
        np.testing.assert_allclose(c[0], a[0], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(c[1], a[1], rtol=1e-9)


def test_shards_tile_the_frame(pkg, hip):
    """Rows dealt to 3 shards: the union of the shard images is the full frame bit for bit and
    the shard gradients sum to the full gradient."""
    import dataclasses
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(70, 50)
    rp = pkg.RenderParams(spp=6, min_bounces=3, absorb=0.3, seed=4, band_rows=8)
    hip.upload_scene(scene)
    full_img, full_g, full_s = hip.render(cam, rp, backward=True)
    img = np.zeros_like(full_img)
    gsum = np.zeros_like(full_g)
    segs = 0
    for s in range(3):
        rps = dataclasses.replace(rp, shard=s, n_shards=3)
        im_s, g_s, st = hip.render(cam, rps, backward=True)
        rows = pkg.shard_rows(cam.height, 8, 3, s)
        other = np.setdiff1d(np.arange(cam.height), rows)
        assert not im_s[other].any()
        img[rows] = im_s[rows]
        gsum += g_s
        segs += st["segments"]
    np.testing.assert_array_equal(img, full_img)
    np.testing.assert_allclose(gsum, full_g, rtol=1e-9)
    assert segs == full_s["segments"]


def test_matches_oracle_on_random_scenes(pkg, hip, oracle):
    """f64 mode: exact (segment counts, 1e-9).  f32 mode: flip-aware (check_f32_heavy_tailed): seed 11's pixel (6, 28) is black
    in f64 and lit in f32 -- one silhouette-edge path, same segment count."""
    for seed, spp in ((11, 8), (12, 8), (13, 8), (12, 32)):
        scene = pkg.random_scene(seed)
        cam = pkg.Camera(48, 40).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
        rp = pkg.RenderParams(spp=spp, min_bounces=2, absorb=0.35, seed=seed)
        adj = np.random.RandomState(seed).uniform(0, 1, (40, 48, 3)).astype(np.float32)
        ref = oracle.render(scene, cam, rp, backward=True, adjoint=adj)
        hip.upload_scene(scene)
        img, grads, stats = hip.render(cam, rp, backward=True, adjoint=adj, f64=True)
        assert stats["segments"] == ref["stats"]["segments"]
        assert grad_rel_err(grads, ref["grads"]) < 1e-9
        np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
        check_f32_heavy_tailed(hip, cam, rp, adj, scene.n_params, ref["image"], ref["grads"], ref["stats"]["segments"],
                               lambda p: oracle.render(scene, cam, rp, backward=True, adjoint=adj, grad_image_param=p)["grad_image"])


def test_matches_the_reference_binary_live(pkg, hip, oracle):
    """Where oracle/_ref/ref_harness travelled with the snapshot (the unmodified reference headers behind our driver,
    compiled in the build container; nothing of /root/reference is read here): fresh scenes that are in no fixture, the
    device against the REFERENCE'S OWN forward and backward() -- f64 mode to 1e-9 with identical segment counts, f32 mode
    flip-aware -- for both integration operators of integrate.hpp."""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/ref_harness is not here")
    for seed, unbiased in ((301, False), (302, False), (303, True)):
        scene = pkg.random_scene(seed) if seed != 302 else pkg.cornell_box(front_specular=True)
        cam = pkg.Camera(30, 22).look_at((0.05, 0.1, -0.15), (0, 0.15, 1)) if seed != 302 else pkg.cornell_camera(30, 22)
        rp = pkg.RenderParams(spp=4, min_bounces=2, absorb=0.3, seed=seed)
        adj = np.random.RandomState(seed).uniform(0.1, 1, (22, 30, 3)).astype(np.float32)
        ref = oracle.render_reference(scene, cam, rp, backward=True, adjoint=adj, tracer_mode=2 if unbiased else 0,
                                      zero_dir_miss=unbiased)           # (as oracle/gen_golden.py runs the two operators)
        hip.upload_scene(scene)
        img, grads, stats = hip.render(cam, rp, backward=True, adjoint=adj, f64=True, unbiased=unbiased)
        assert stats["segments"] == ref["stats"]["segments"]
        assert grad_rel_err(grads, ref["grads"]) < 1e-9
        np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
        if not unbiased:
            check_f32_heavy_tailed(hip, cam, rp, adj, scene.n_params, ref["image"], ref["grads"], ref["stats"]["segments"],
                                   lambda p: oracle.render(scene, cam, rp, backward=True, adjoint=adj, grad_image_param=p)["grad_image"])


def test_edge_cases(pkg, hip, oracle):
    scene = pkg.cornell_box()
    hip.upload_scene(scene)
    # 1x1 image, 1 spp; ragged sizes; every path absorbed at depth 0 (min_bounces 0, absorb 1)
    for (w, h, spp, b, p) in [(1, 1, 1, 1, 0.5), (33, 17, 3, 0, 0.25), (5, 7, 2, 0, 1.0), (16, 16, 1, 2, 0.0)]:
        cam = pkg.cornell_camera(w, h)
        rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=3, max_depth=12)
        ref = oracle.render(scene, cam, rp, backward=True)
        img, grads, stats = hip.render(cam, rp, backward=True, f64=True)
        assert stats["segments"] == ref["stats"]["segments"]
        np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
        np.testing.assert_allclose(grads, ref["grads"], rtol=1e-9, atol=1e-12)
    # a parameter that does not require grad reports a zero gradient
    scene2 = pkg.cornell_box()
    scene2.requires_grad[2] = False
    hip.upload_scene(scene2)
    _, grads, _ = hip.render(pkg.cornell_camera(16, 16), pkg.RenderParams(spp=2, min_bounces=3, absorb=1.0), backward=True)
    assert not grads[2].any() and grads[0].any()
    # update_params == re-upload
    hip.upload_scene(scene)
    newp = np.array(scene.params) * 0.5 + 0.1
    hip.update_params(newp)
    cam = pkg.cornell_camera(24, 24)
    rp = pkg.RenderParams(spp=4, min_bounces=3, absorb=1.0, seed=8)
    a = hip.render(cam, rp, backward=True)
    scene3 = pkg.cornell_box()
    scene3.params = [tuple(v) for v in newp]
    hip.upload_scene(scene3)
    b = hip.render(cam, rp, backward=True)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_error_behaviour(pkg):
    r = pkg.HipRenderer(0)
    cam = pkg.cornell_camera(8, 8)
    r.scene = pkg.cornell_box()          # python-side only: the context has no scene yet
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_NO_SCENE"):
        r.render(cam, pkg.RenderParams(spp=1))
    r.upload_scene(pkg.cornell_box())
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        r.render(cam, pkg.RenderParams(spp=0))
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        r.render(cam, pkg.RenderParams(spp=1, absorb=1.5))
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):      # deeper than DRT_MAX_DEPTH: refused, not truncated
        r.render(cam, pkg.RenderParams(spp=1, max_depth=pkg.MAX_DEPTH + 1))
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        r.render(cam, pkg.RenderParams(spp=1, min_bounces=pkg.MAX_DEPTH + 6, absorb=1.0))
    bad = pkg.cornell_box()
    bad.shapes[0] = (pkg.SHAPE_SPHERE, 99, -1, (0., 0., 3., 1.))
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        r.upload_scene(bad)
    with pytest.raises(pkg.DrtHipError):
        pkg.HipRenderer(4096)            # no such device
    r.close()


def test_capped_paths_are_reported(pkg, hip, oracle):
    """The reference ends paths by roulette only; max_depth is this library's extension.  Paths it cuts short are
    counted (drt_hip_stats.capped_paths), so the bias of a cap is visible; a cap that IS the roulette's certain
    kill (absorb == 1 at min_bounces) cuts nothing."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(64, 48)
    hip.upload_scene(scene)
    _, _, st = hip.render(cam, pkg.RenderParams(spp=4, min_bounces=5, absorb=1.0, seed=2), backward=True)
    assert st["capped_paths"] == 0
    for nb in (0, 1, 3):
        rp = pkg.RenderParams(spp=4, min_bounces=1, absorb=0.1, seed=2, max_depth=6, bounces_per_launch=nb)
        _, _, st = hip.render(cam, rp, backward=True, f64=True)
        # the oracle run without the cap tells which paths reach vertex 7: those with > 6 vertices
        ref6 = oracle.render(scene, cam, rp, backward=False)
        ref_long = oracle.render(scene, cam, dataclasses_replace(rp, max_depth=7), backward=False)
        assert st["segments"] == ref6["stats"]["segments"]
        assert st["capped_paths"] == ref_long["stats"]["segments"] - ref6["stats"]["segments"] > 0
    _, _, st = hip.render(cam, pkg.RenderParams(spp=4, min_bounces=1, absorb=0.3, seed=2), backward=True)
    assert st["capped_paths"] == 0       # the default cap of 64 vertices is never reached at this absorb (0.7^63)


def dataclasses_replace(obj, **kw):
    import dataclasses
    return dataclasses.replace(obj, **kw)


def test_finite_difference_of_albedo(pkg, hip):
    """Sampling never depends on albedo values, so central differences of the render at a fixed
    seed reproduce the reverse-mode gradient.  Done in the f64 device mode."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(48, 48)
    rp = pkg.RenderParams(spp=8, min_bounces=4, absorb=1.0, seed=2)
    hip.upload_scene(scene)
    _, grads, _ = hip.render(cam, rp, backward=True, f64=True)
    base = np.array(scene.params, dtype=np.float64)
    # gradient of sum over paths of radiance: compare with d/dp of sum(image) * spp via the
    # oracle-free identity using two f32-returning renders is too coarse, so use update_params
    # with a large, exactly representable step: radiance is a polynomial in each albedo.
    h = 2.0 ** -6
    for p, c in [(0, 0), (2, 1)]:
        vals = []
        for sgn in (+1, -1):
            q = base.copy()
            q[p, c] += sgn * h
            hip.update_params(q)
            img, _, _ = hip.render(cam, rp, backward=False, f64=True)
            vals.append(img.astype(np.float64).sum((0, 1))[c] * rp.spp)
        fd = (vals[0] - vals[1]) / (2 * h)
        # polynomial of degree <= 4 in this albedo: central difference error is O(h^2 f''')
        assert abs(fd - grads[p, c]) <= 5e-3 * abs(grads[p, c])
    hip.update_params(base)


@pytest.mark.parametrize("name", ["g12_gradimage_red_48x36x8_d4", "g13_gradimage_white_40x40x6_rr",
                                  "m4_mirror_gradimage_white_32x32x6"])
def test_gradient_image_matches_reference(pkg, hip, name):
    """drt_hip_render_gradient_image against the per-pixel gradients the reference produces."""
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    p = g["case"]["grad_image_param"]
    hip.upload_scene(scene)
    img, gimg, st = hip.render_gradient_image(cam, rp, p, adjoint=adjoint, f64=True)
    assert st["segments"] == int(g["segments"])
    scale = np.abs(g["grad_image"]).max()
    np.testing.assert_allclose(gimg, g["grad_image"].astype(np.float32), rtol=2e-7, atol=1e-7 * scale)
    np.testing.assert_allclose(img, g["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    img, gimg, st = hip.render_gradient_image(cam, rp, p, adjoint=adjoint)
    bad = np.abs(gimg.astype(np.float64) - g["grad_image"]).max(-1) > PIXEL_TOL * scale
    assert bad.sum() <= flip_budget(cam.width * cam.height * rp.spp)
    # its pixel sum is the ordinary gradient of that parameter
    _, grads, _ = hip.render(cam, rp, backward=True, adjoint=adjoint)
    np.testing.assert_allclose(gimg.astype(np.float64).sum((0, 1)) * rp.spp, grads[p], rtol=1e-5)
    with pytest.raises(pkg.DrtHipError, match="DRT_ERR_INVALID"):
        hip.render_gradient_image(cam, rp, 99)


from conftest import UNBIASED_GOLDENS  # noqa: E402


@pytest.mark.parametrize("name", UNBIASED_GOLDENS)
def test_unbiased_backward_matches_reference(pkg, hip, name):
    """DRT_RENDER_UNBIASED: the adjoint-round wavefront against fixtures produced by the reference's
    own integrate(..., unbiased=true) (harness tracer): same image as the biased mode, gradients from
    fresh samples at every vertex, O(depth^2) segments -- all identical in the f64 device mode."""
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True, adjoint=adjoint, f64=True, unbiased=True)
    assert st["segments"] == int(g["segments"])
    assert grad_rel_err(grads, g["grads"]) < 1e-9
    np.testing.assert_allclose(img, g["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    # analytic scenes: the whole operator in one launch (k_path_unbiased); a mesh: the adjoint-round wavefront.  The
    # wavefront on request (bounces_per_launch = 1) gives the same numbers
    assert st["kernels"]["path"]["launches"] == (0 if "mesh" in name else 1)
    import dataclasses
    _, gq, sq = hip.render(cam, dataclasses.replace(rp, bounces_per_launch=1), backward=True, adjoint=adjoint, f64=True, unbiased=True)
    assert sq["kernels"]["path"]["launches"] == 0 and sq["segments"] == st["segments"] and grad_rel_err(gq, grads) < 1e-9
    img, grads, st = hip.render(cam, rp, backward=True, adjoint=adjoint, unbiased=True)
    assert abs(st["segments"] - int(g["segments"])) <= 64
    # measured on the six fixtures (profiles/r02_parity_report.txt): <= 3.5e-6, no flipped path; the bound is 3 x that
    assert grad_rel_err(grads, g["grads"]) <= 1e-5
    # deterministic
    img2, grads2, _ = hip.render(cam, rp, backward=True, adjoint=adjoint, unbiased=True)
    np.testing.assert_array_equal(grads, grads2)


from conftest import DEPTH_LIMIT_GOLDENS  # noqa: E402


@pytest.mark.parametrize("name", DEPTH_LIMIT_GOLDENS)
def test_unbiased_chains_that_end_at_the_depth_limit_stay_in_step_with_the_reference(pkg, hip, name):
    """The library ends paths at 64 vertices (DRT_MAX_DEPTH); the reference has no limit.  Where the reference's own roulette
    ends a path exactly THERE -- depth 64: its trace() draws, and the draw kills -- nothing is cut short (capped_paths == 0)
    and the device must consume that draw too, or every later suffix of the chain draws other numbers (found by the long
    fuzz of round 4: 20 rays of one path in 92,667).  Fixtures from the reference's own integrate(..., unbiased=true)."""
    import dataclasses
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    hip.upload_scene(scene)
    for nb in (0, 1):                                      # the one-launch kernel (analytic scenes) and the adjoint-round wavefront
        _, grads, st = hip.render(cam, dataclasses.replace(rp, bounces_per_launch=nb), backward=True, adjoint=adjoint, f64=True, unbiased=True)
        assert st["capped_paths"] == 0
        assert st["segments"] == int(g["segments"])
        assert grad_rel_err(grads, g["grads"]) < 1e-9
        assert st["kernels"]["path"]["launches"] == (1 if nb == 0 and "mesh" not in name else 0)
    # f32: chains this long are chaotic (a rounding difference grows with every bounce off a sphere): finite, and close
    _, grads, st = hip.render(cam, rp, backward=True, adjoint=adjoint, unbiased=True)
    assert np.isfinite(grads).all() and abs(st["segments"] - int(g["segments"])) <= 0.02 * int(g["segments"])
    assert grad_rel_err(grads, g["grads"]) <= 1e-3


def test_a_cap_where_the_roulette_ends_every_path_anyway_changes_nothing(pkg, hip):
    """max_depth at or beyond min_bounces with absorb == 1: the roulette ends every path there and its draw is consumed, as in
    the reference (no cap) -- the unbiased operator's later suffixes draw the same numbers, capped or not, in both routes."""
    import dataclasses
    scene = pkg.scene_by_name("cornell_mirror_wall")
    cam = pkg.cornell_camera(24, 18)
    hip.upload_scene(scene)
    for nb in (0, 1):
        rp = pkg.RenderParams(spp=3, min_bounces=3, absorb=1.0, seed=294368374, bounces_per_launch=nb)
        _, g0, s0 = hip.render(cam, rp, backward=True, f64=True, unbiased=True)
        for cap in (3, 5):
            _, g1, s1 = hip.render(cam, dataclasses.replace(rp, max_depth=cap), backward=True, f64=True, unbiased=True)
            assert s1["segments"] == s0["segments"] and s1["capped_paths"] == 0
            np.testing.assert_array_equal(g1, g0)


def test_absorb_one_ends_every_path_even_where_the_references_draw_is_exactly_one(pkg, hip):
    """The one deliberate deviation in a result: rand() == RAND_MAX makes the reference's uniform() 1.0, `1.0 < absorb` fails for
    absorb == 1, the path goes on with survival probability 0 and is divided by it -- NaN in the reference's image and
    gradients (tests/test_oracle_properties.py pins that).  The device ends the path: finite, one ray fewer, and every other
    pixel the reference's."""
    g = load_golden("q1_nan_mirror_wall_15x36x10_d5")       # from the reference itself: NaN at pixel (3, 14), NaN gradients
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    assert not np.isfinite(g["image"][14, 3]).any() and not np.isfinite(g["grads"]).all()
    hip.upload_scene(scene)
    for kw in (dict(f64=True), dict()):
        img, grads, st = hip.render(cam, rp, backward=True, adjoint=adjoint, **kw)
        assert np.isfinite(img).all() and np.isfinite(grads).all() and st["capped_paths"] == 0
        assert st["segments"] == int(g["segments"]) - 1
        ok = np.ones((36, 15), bool); ok[14, 3] = False
        np.testing.assert_allclose(img[ok], g["image"][ok].astype(np.float32), rtol=2e-7 if kw else 2e-4, atol=1e-12 if kw else 1e-6)
        # the parameters the path does not touch keep the reference's gradients
        fin = np.isfinite(g["grads"]).all(1)
        assert fin.any() and grad_rel_err(grads[fin], g["grads"][fin]) < (1e-9 if kw else 1e-4)


def test_unbiased_and_biased_gradients_agree_statistically(pkg, hip):
    """Two estimators of the same derivative: at 256x256x16 they agree within Monte-Carlo noise."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(256, 256)
    rp = pkg.RenderParams(spp=16, min_bounces=5, absorb=1.0, seed=12)
    hip.upload_scene(scene)
    _, gb, sb = hip.render(cam, rp, backward=True)
    _, gu, su = hip.render(cam, rp, backward=True, unbiased=True)
    assert su["segments"] > 2 * sb["segments"]
    np.testing.assert_allclose(gu, gb, rtol=0.03)


def test_f32_half_vector_singularity_regression(pkg, hip):
    """Row 848 of a 1024x1024x32 depth-12 render with the specular sphere: sample 21 of pixel
    (976, 848) reflects almost straight through (h . wi ~ 0); the literal normalize(wi + wo) of
    bxdf.hpp:97 cancelled to normalize(0) = NaN in f32 and poisoned the pixel and every gradient."""
    scene = pkg.cornell_box(front_specular=True)
    cam = pkg.cornell_camera(1024, 1024)
    rp = pkg.RenderParams(spp=32, min_bounces=12, absorb=1.0, seed=3, shard=848, n_shards=1024, band_rows=1)
    hip.upload_scene(scene)
    img, grads, _ = hip.render(cam, rp, backward=True)
    assert np.isfinite(img).all() and np.isfinite(grads).all()
    img64, grads64, _ = hip.render(cam, rp, backward=True, f64=True)
    np.testing.assert_allclose(img[848, 976], img64[848, 976], rtol=2e-3, atol=1e-6)
    assert grad_rel_err(grads, grads64) < 1e-4


def test_malformed_scenes_are_rejected_not_crashed(pkg):
    """Random corruptions of a valid scene description: the library must answer with
    DRT_ERR_INVALID / DRT_ERR_UNSUPPORTED (or accept a still-valid scene), never crash, and stay
    usable afterwards."""
    r = pkg.HipRenderer(0)
    rs = np.random.RandomState(7)
    cam = pkg.cornell_camera(8, 8)
    for trial in range(60):
        sc = pkg.random_scene(trial % 5) if trial % 3 else pkg.cornell_with_mesh(6, 8, per_face_params=3)
        kind = rs.randint(7)
        if kind == 0:
            i = rs.randint(len(sc.shapes)); t, m, e, p = sc.shapes[i]; sc.shapes[i] = (t, int(rs.randint(-5, 200)), e, p)
        elif kind == 1:
            i = rs.randint(len(sc.shapes)); t, m, e, p = sc.shapes[i]; sc.shapes[i] = (t, m, int(rs.randint(-5, 200)), p)
        elif kind == 2:
            i = rs.randint(len(sc.shapes)); t, m, e, p = sc.shapes[i]; sc.shapes[i] = (int(rs.randint(3, 9)), m, e, p)
        elif kind == 3:
            i = rs.randint(len(sc.materials)); t, p, ex = sc.materials[i]; sc.materials[i] = (t, int(rs.randint(-3, 500)), ex)
        elif kind == 4:
            i = rs.randint(len(sc.materials)); t, p, ex = sc.materials[i]; sc.materials[i] = (int(rs.randint(2, 6)), p, ex)
        elif kind == 5 and sc.emitters:
            sc.emitters[rs.randint(len(sc.emitters))] = int(rs.randint(-3, 500))
        elif kind == 6 and sc.meshes:
            v, idx, fm = sc.meshes[0]; idx = idx.copy(); idx[rs.randint(len(idx)), rs.randint(3)] = 10 ** 6; sc.meshes[0] = (v, idx, fm)
        try:
            r.upload_scene(sc)
            img, _, _ = r.render(cam, pkg.RenderParams(spp=1, min_bounces=2, absorb=0.5), backward=True)
            assert img.shape == (8, 8, 3)
        except pkg.DrtHipError as ex:
            assert "DRT_ERR_INVALID" in str(ex) or "DRT_ERR_UNSUPPORTED" in str(ex)
    r.upload_scene(pkg.cornell_box())
    img, g, _ = r.render(cam, pkg.RenderParams(spp=2, min_bounces=2, absorb=0.5), backward=True)
    assert np.isfinite(img).all() and np.isfinite(g).all()
    r.close()


def test_mirror_bxdf_at_scale_and_update_params(pkg, hip, oracle):
    """MirrorBxDF (bxdf.hpp:126-144 repaired): a larger frame against the oracle in both device modes, a
    mirror next to more than 4 parameters (general gradient path + the internal constant), and
    update_params leaving the internal constant alone."""
    scene = pkg.scene_by_name("cornell_mirror_wall")
    cam = pkg.cornell_camera(96, 80)
    rp = pkg.RenderParams(spp=8, min_bounces=3, absorb=0.25, seed=23)
    ref = oracle.render(scene, cam, rp, backward=True)
    hip.upload_scene(scene)
    img, grads, stats = hip.render(cam, rp, backward=True, f64=True)
    assert stats["segments"] == ref["stats"]["segments"]
    assert grads.shape == (scene.n_params, 3) and grad_rel_err(grads, ref["grads"]) < 1e-9
    np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    img, grads, stats = hip.render(cam, rp, backward=True)
    check_f32(img, grads, stats["segments"], ref["image"], ref["grads"], ref["stats"]["segments"])
    # update_params == re-upload (the internal constant of the mirror must survive the update)
    newp = np.array(scene.params) * 0.7 + 0.05
    hip.update_params(newp)
    a = hip.render(cam, rp, backward=True)
    scene.params = [tuple(v) for v in newp]
    hip.upload_scene(scene)
    b = hip.render(cam, rp, backward=True)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    # a mirror sphere among random spheres with 9+ parameters
    scene = pkg.random_scene(4)
    scene.sphere((0.3, -1.2, 2.2), 0.8, scene.mirror())
    cam = pkg.Camera(48, 40).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
    rp = pkg.RenderParams(spp=8, min_bounces=2, absorb=0.35, seed=4)
    ref = oracle.render(scene, cam, rp, backward=True)
    hip.upload_scene(scene)
    img, grads, stats = hip.render(cam, rp, backward=True, f64=True)
    assert stats["segments"] == ref["stats"]["segments"] and grad_rel_err(grads, ref["grads"]) < 1e-9
    ref_u = oracle.render(scene, cam, rp, backward=True, unbiased=True, zero_dir_miss=True)
    img, grads, stats = hip.render(cam, rp, backward=True, f64=True, unbiased=True)
    assert stats["segments"] == ref_u["stats"]["segments"] and grad_rel_err(grads, ref_u["grads"]) < 1e-9


def test_config5_shape_properties_at_scale(pkg, hip):
    """BASELINE config 5's shape (specular front sphere, depth 16) on a 1024 x 1024 frame, through
    size-independent properties: linearity in the emission, bitwise determinism, several batches vs
    one, and the f32 kernels against the device's own flip-free f64 mode (same paths)."""
    import dataclasses
    scene = pkg.cornell_box(front_specular=True)
    cam = pkg.cornell_camera(1024, 1024)
    rp = pkg.RenderParams(spp=8, min_bounces=16, absorb=1.0, seed=2)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True)
    assert np.isfinite(img).all() and np.isfinite(grads).all()
    assert st["paths"] == 1024 * 1024 * 8 and st["segments"] > 10 * st["paths"]
    e = scene.param_names.index("emission")
    total = img.astype(np.float64).sum((0, 1)) * rp.spp
    np.testing.assert_allclose(grads[e] * np.array(scene.params[e]), total, rtol=5e-6)
    again = hip.render(cam, rp, backward=True)
    np.testing.assert_array_equal(again[0], img)
    np.testing.assert_array_equal(again[1], grads)
    split = hip.render(cam, dataclasses.replace(rp, batch_paths=3_000_000), backward=True)
    assert split[2]["batches"] > 1 and split[2]["segments"] == st["segments"]
    np.testing.assert_allclose(split[1], grads, rtol=1e-7)
    img64, g64, st64 = hip.render(cam, rp, backward=True, f64=True)
    assert abs(st64["segments"] - st["segments"]) <= 2e-5 * st64["segments"]
    assert grad_rel_err(grads, g64) < GRAD_TOL
    m32, m64 = img.astype(np.float64).mean((0, 1)), img64.astype(np.float64).mean((0, 1))
    assert np.abs(m32 - m64).max() <= MEAN_TOL * m64.max()


@pytest.mark.parametrize("scene_name,b,p,unbiased", [("cornell", 8, 1.0, False), ("cornell", 1, 0.4, False),
                                                     ("cornell_specular", 3, 0.2, False), ("cornell_mirror_wall", 5, 1.0, False),
                                                     ("cornell", 4, 1.0, True), ("cornell_specular", 2, 0.3, True)])
def test_results_do_not_depend_on_bounces_per_launch(pkg, hip, scene_name, b, p, unbiased):
    """A shade launch may carry its rays through 1..8 bounces in registers (drt_render_params.
    bounces_per_launch): image, gradients and segment count are bitwise the same.  0 = the library's choice, which
    for fixed-depth renders of analytic scenes is k_path (the whole path in one launch, gradients accumulated in
    path order instead of by a reverse tape walk): the same paths and the same terms, summed in another order --
    equal segment counts, values equal to f32 rounding."""
    import dataclasses
    hip.upload_scene(pkg.scene_by_name(scene_name))
    cam = pkg.cornell_camera(160, 96)
    rp = pkg.RenderParams(spp=6, min_bounces=b, absorb=p, seed=12, max_depth=24)
    ref = hip.render(cam, dataclasses.replace(rp, bounces_per_launch=1), backward=True, unbiased=unbiased)
    for nb in (0, 2, 3, 8):
        got = hip.render(cam, dataclasses.replace(rp, bounces_per_launch=nb), backward=True, unbiased=unbiased)
        if nb == 0 and got[2]["kernels"]["path"]["launches"]:
            # (its f32 closest-hit arithmetic is its own: one flipped path changes the count by its length -- by up to
            #  depth^2 / 2 under the unbiased operator, whose every vertex traces a fresh suffix)
            assert abs(got[2]["segments"] - ref[2]["segments"]) <= (24 * 24 // 2 if unbiased else 64)
            if ("specular" in scene_name and p < 1.0) or "mirror" in scene_name:
                # long chains of glossy bounces (depth up to 24 here), or a mirror facing a glossy sphere, amplify the
                # last-bit differences of the two closest-hit routines: the stated f32 pixel bound instead of rounding
                bad = np.abs(got[0].astype(np.float64) - ref[0]).max(-1) > PIXEL_TOL * float(np.abs(ref[0]).max())
                assert bad.sum() <= flip_budget(160 * 96 * 6)
            else:
                np.testing.assert_allclose(got[0], ref[0], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(got[1], ref[1], rtol=2e-5, atol=1e-6 * np.abs(ref[1]).max())
            continue
        assert got[2]["segments"] == ref[2]["segments"]
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
        overridden = os.environ.get("DRT_HIP_SHADE_BOUNCES")   # (the debug knob wins)
        if nb == 8 and not unbiased and not overridden:
            assert got[2]["kernels"]["shade"]["launches"] < ref[2]["kernels"]["shade"]["launches"]
            assert got[2]["queue_rays_read"] < ref[2]["queue_rays_read"]


@pytest.mark.parametrize("scene_name,unbiased", [("cornell", False), ("cornell_specular", False), ("cornell", True)])
def test_the_textbook_wavefront_agrees_with_the_fused_one(pkg, hip, scene_name, unbiased):
    """DRT_RENDER_UNFUSED: K1, then per bounce K2 (k_intersect) and K3 (k_shade) as launches of their own -- the
    pipeline as SURVEY section 8 lists it -- against the fused one-launch-per-bounce route: the same paths (segment
    count), values equal to f32 rounding (K2's closest-hit loop and the fused launch's packed test differ in the last bit)."""
    import dataclasses
    hip.upload_scene(pkg.scene_by_name(scene_name))
    cam = pkg.cornell_camera(96, 80)
    rp = pkg.RenderParams(spp=5, min_bounces=6, absorb=1.0, seed=21, bounces_per_launch=1)
    ref = hip.render(cam, rp, backward=True, unbiased=unbiased)
    got = hip.render(cam, dataclasses.replace(rp, flags=rp.flags | pkg.RENDER_UNFUSED), backward=True, unbiased=unbiased)
    assert ref[2]["kernels"]["intersect"]["launches"] == 0 and got[2]["kernels"]["intersect"]["launches"] >= 6
    assert got[2]["kernels"]["raygen"]["launches"] == 1
    assert got[2]["segments"] == ref[2]["segments"]
    if "specular" in scene_name:
        # (glossy bounces amplify the last-bit differences of the two closest-hit routines: the stated f32 pixel bound)
        bad = np.abs(got[0].astype(np.float64) - ref[0]).max(-1) > PIXEL_TOL * float(np.abs(ref[0]).max())
        assert bad.sum() <= flip_budget(96 * 80 * 5)
    else:
        np.testing.assert_allclose(got[0], ref[0], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(got[1], ref[1], rtol=2e-5, atol=1e-6 * np.abs(ref[1]).max())


@pytest.mark.parametrize("f64", [False, True])
def test_forward_only_image_equals_the_image_of_a_gradient_render(pkg, hip, f64):
    """Forward-only renders take their own route (for depth caps <= 8 the shade launch walks every path's
    tape in place, without a tape in HBM): the image must be the one a forward + backward render returns."""
    for scene_name, b, p in (("cornell", 8, 1.0), ("cornell_specular", 5, 1.0), ("cornell", 2, 0.3), ("mesh10x12", 4, 1.0)):
        hip.upload_scene(pkg.scene_by_name(scene_name))
        cam = pkg.cornell_camera(144, 100)
        rp = pkg.RenderParams(spp=5, min_bounces=b, absorb=p, seed=31)
        fwd, g0, st0 = hip.render(cam, rp, backward=False, f64=f64)
        both, g1, st1 = hip.render(cam, rp, backward=True, f64=f64)
        assert g0 is None and st0["segments"] == st1["segments"]
        np.testing.assert_array_equal(fwd, both)


@pytest.mark.parametrize("scene_name,b,p,max_depth,spp", [("cornell", 1, 0.5, 0, 7), ("cornell", 0, 0.3, 0, 5), ("cornell", 1, 0.5, 5, 6),
                                                           ("cornell_specular", 3, 0.2, 0, 5), ("cornell_specular", 2, 0.35, 12, 4)])
def test_path_kernel_regenerating_lanes(pkg, hip, oracle, scene_name, b, p, max_depth, spp):
    """Roulette-terminated renders (the reference's defaults are -b 1 -p 0.5) take k_path's regenerating form: every
    lane on its own depth, restarting with its next sample when its path ends.  Against the oracle: f64 mode to 1e-9
    (same paths, same segment and capped-path counts), f32 mode within the stated bounds; frame width not a multiple
    of the wave, sample ranges that do not divide the samples, several batches; forward-only gives the same image."""
    import dataclasses
    scene = pkg.scene_by_name(scene_name)
    cam = pkg.cornell_camera(77, 45)
    rp = pkg.RenderParams(spp=spp, min_bounces=b, absorb=p, seed=3, **({"max_depth": max_depth} if max_depth else {}))
    ref = oracle.render(scene, cam, rp, backward=True)
    hip.upload_scene(scene)
    img, grads, st = hip.render(cam, rp, backward=True, f64=True)
    assert st["kernels"]["path"]["launches"] == 1 and st["kernels"]["shade"]["launches"] == 0
    assert st["segments"] == ref["stats"]["segments"]
    assert st["capped_paths"] == ref["stats"].get("capped_paths", st["capped_paths"])
    if max_depth in (0, 5):
        assert (st["capped_paths"] > 0) == (max_depth == 5)
    assert grad_rel_err(grads, ref["grads"]) < 1e-9
    np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    for batch in (77 * 45 * 2 + 5, 999):
        many = hip.render(cam, dataclasses.replace(rp, batch_paths=batch), backward=True, f64=True)
        assert many[2]["batches"] > 1 and many[2]["segments"] == st["segments"]
        np.testing.assert_allclose(many[0], img, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(many[1], grads, rtol=1e-12)
    fwd, _, fst = hip.render(cam, rp, backward=False, f64=True)
    assert fst["segments"] == st["segments"]
    np.testing.assert_array_equal(fwd, img)
    img32, g32, st32 = hip.render(cam, rp, backward=True)
    assert st32["kernels"]["path"]["launches"] == 1
    check_f32(img32, g32, st32["segments"], ref["image"], ref["grads"], ref["stats"]["segments"], n_paths=77 * 45 * spp)


def test_path_kernel_corner_cases(pkg, hip, oracle):
    """k_path (the one-launch path kernel) beyond the fixtures: several batches (pixels and samples both cut) equal one
    batch; roulette at depth 0 under a small max_depth; a forward-only render of a scene with more than 4 parameters
    (no tangents, parameters from LDS or memory) and its run-time intersection program -- against the oracle."""
    import dataclasses
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(100, 60)
    rp = pkg.RenderParams(spp=12, min_bounces=5, absorb=1.0, seed=21)
    hip.upload_scene(scene)
    one = hip.render(cam, rp, backward=True)
    assert one[2]["kernels"]["path"]["launches"] == 1 and one[2]["batches"] == 1
    for batch in (100 * 60 * 5 + 7, 1234):
        many = hip.render(cam, dataclasses.replace(rp, batch_paths=batch), backward=True)
        assert many[2]["batches"] > 1 and many[2]["kernels"]["path"]["launches"] == many[2]["batches"]
        assert many[2]["segments"] == one[2]["segments"]
        np.testing.assert_allclose(many[0], one[0], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(many[1], one[1], rtol=1e-7)
    # roulette from depth 0 on, paths cut at 4 vertices: k_path's depth-0 roulette and its capped-path count
    rp0 = pkg.RenderParams(spp=6, min_bounces=0, absorb=0.15, seed=5, max_depth=4)
    ref = oracle.render(scene, cam, rp0, backward=True)
    img, grads, st = hip.render(cam, rp0, backward=True, f64=True)
    assert st["kernels"]["path"]["launches"] == 1 and st["segments"] == ref["stats"]["segments"] and st["capped_paths"] > 0
    assert grad_rel_err(grads, ref["grads"]) < 1e-9
    np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    img, grads, st = hip.render(cam, rp0, backward=True)
    check_f32(img, grads, st["segments"], ref["image"], ref["grads"], ref["stats"]["segments"], n_paths=100 * 60 * 6)
    # 5 .. 8 parameters: k_path with eight tangents (run-time intersection program: non-axis planes, lights with a BxDF)
    rscene = pkg.random_scene(3, specular=False)
    assert 4 < rscene.n_params <= 8
    rcam = pkg.Camera(64, 48).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
    rrp = pkg.RenderParams(spp=6, min_bounces=5, absorb=1.0, seed=17)
    rref = oracle.render(rscene, rcam, rrp, backward=True)
    hip.upload_scene(rscene)
    fwd, _, fst = hip.render(rcam, rrp, backward=False, f64=True)
    assert fst["kernels"]["path"]["launches"] == 1 and fst["segments"] == rref["stats"]["segments"]
    np.testing.assert_allclose(fwd, rref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
    both, g, bst = hip.render(rcam, rrp, backward=True, f64=True)
    assert bst["kernels"]["path"]["launches"] == 1 and bst["kernels"]["backward"]["launches"] == 0
    assert grad_rel_err(g, rref["grads"]) < 1e-9
    np.testing.assert_array_equal(both, fwd)
    check_f32_heavy_tailed(hip, rcam, rrp, None, rscene.n_params, rref["image"], rref["grads"], rref["stats"]["segments"],
                           lambda p: oracle.render(rscene, rcam, rrp, backward=True, grad_image_param=p)["grad_image"])
    # more than 8 parameters: forward-only goes through k_path without tangents, backward through its general form (vertex
    # history + per-wave tables, tests/test_gpu_many_params.py) -- and through the tape where a launch per bounce is asked for
    import dataclasses
    bscene = pkg.random_scene(3, specular=False, n_lights=6)
    assert bscene.n_params > 8
    bref = oracle.render(bscene, rcam, rrp, backward=True)
    hip.upload_scene(bscene)
    fwd, _, fst = hip.render(rcam, rrp, backward=False, f64=True)
    assert fst["kernels"]["path"]["launches"] == 1 and fst["segments"] == bref["stats"]["segments"]
    both, g, bst = hip.render(rcam, rrp, backward=True, f64=True)
    assert bst["kernels"]["path"]["launches"] == 1 and bst["kernels"]["backward"]["launches"] == 0
    assert grad_rel_err(g, bref["grads"]) < 1e-9
    np.testing.assert_array_equal(both, fwd)
    _, gq, qst = hip.render(rcam, dataclasses.replace(rrp, bounces_per_launch=1), backward=True, f64=True)
    assert qst["kernels"]["path"]["launches"] == 0 and qst["kernels"]["backward"]["launches"] > 0 and grad_rel_err(gq, bref["grads"]) < 1e-9


def test_path_kernel_scene_with_many_shapes(pkg, hip, oracle):
    """k_path's kind-sorted LDS program covers every analytic scene the ABI takes (64 shapes), not only the 16 whose
    kinds fit the compiled-in signature: 31 shapes (24 random spheres, 6 walls, a light), glossy materials, roulette."""
    scene = pkg.random_scene(11, n_spheres=24)
    assert len(scene.shapes) == 31 and scene.n_params <= 8
    cam = pkg.Camera(72, 48).look_at((0.1, 0.0, -0.2), (0, 0.2, 1))
    hip.upload_scene(scene)
    for b, p in ((5, 1.0), (2, 0.3)):
        rp = pkg.RenderParams(spp=5, min_bounces=b, absorb=p, seed=23)
        ref = oracle.render(scene, cam, rp, backward=True)
        img, g, st = hip.render(cam, rp, backward=True, f64=True)
        assert st["kernels"]["path"]["launches"] == 1 and st["segments"] == ref["stats"]["segments"]
        assert grad_rel_err(g, ref["grads"]) < 1e-9
        np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
        img32, g32, st32 = hip.render(cam, rp, backward=True)
        assert st32["kernels"]["path"]["launches"] == 1
        check_f32_heavy_tailed(hip, cam, rp, None, scene.n_params, ref["image"], ref["grads"], ref["stats"]["segments"],
                               lambda p: oracle.render(scene, cam, rp, backward=True, grad_image_param=p)["grad_image"])


def test_eight_parameters_every_walls_albedo(pkg, hip, oracle):
    """The reference's scene with an albedo parameter per wall (8 parameters): k_path carries eight tangents; against
    the oracle in both modes, and the gradient of the shared parameters equals the 4-parameter scene's where the
    walls that got their own parameter keep white's value."""
    scene = pkg.scene_by_name("cornell_walls")
    assert scene.n_params == 8
    cam = pkg.cornell_camera(96, 64)
    for b, p in ((6, 1.0), (2, 0.4)):
        rp = pkg.RenderParams(spp=6, min_bounces=b, absorb=p, seed=8)
        ref = oracle.render(scene, cam, rp, backward=True)
        hip.upload_scene(scene)
        img, g, st = hip.render(cam, rp, backward=True, f64=True)
        assert st["kernels"]["path"]["launches"] == 1 and st["kernels"]["backward"]["launches"] == 0
        assert st["segments"] == ref["stats"]["segments"] and grad_rel_err(g, ref["grads"]) < 1e-9
        np.testing.assert_allclose(img, ref["image"].astype(np.float32), rtol=2e-7, atol=1e-12)
        img32, g32, st32 = hip.render(cam, rp, backward=True)
        check_f32(img32, g32, st32["segments"], ref["image"], ref["grads"], ref["stats"]["segments"], n_paths=96 * 64 * 6)
        # the queue wavefront (tape + K6) gives the same gradients
        _, gq, stq = hip.render(cam, dataclasses_replace(rp, bounces_per_launch=1), backward=True, f64=True)
        assert stq["kernels"]["path"]["launches"] == 0 and grad_rel_err(gq, g) < 1e-9


def tie_scene(pkg, general_first=True):
    """The Cornell box with its back wall doubled: the SAME plane z = 6 once as a general plane record (normal (0, 0, -2),
    offset -12) and once as an axis plane (normal (0, 0, -1), offset -6).  Both forms give bit-identical t in f32 and in
    f64 (power-of-two scalings are exact; v_rcp is exact under them), so every ray that reaches the wall is an exact tie
    between shapes of two different KINDS of the device's intersection program -- which the reference resolves by scene
    order (pathtracer.hpp:80: `t >= tmin` skips).  The two copies carry different albedo parameters."""
    s = pkg.cornell_box()
    first = s.diffuse(s.parameter((0.7, 0.3, 0.2), True, "wall_first"))
    second = s.diffuse(s.parameter((0.2, 0.3, 0.7), True, "wall_second"))
    general = (pkg.SHAPE_PLANE, first if general_first else second, -1, (0., 0., -2., -12.))
    axis = (pkg.SHAPE_PLANE, second if general_first else first, -1, (0., 0., -1., -6.))
    pair = [general, axis] if general_first else [axis, general]
    s.shapes[4:5] = pair                                   # in place of the back wall (render.cpp:43)
    return s


@pytest.mark.parametrize("general_first", [True, False])
def test_exact_tie_between_kinds_goes_to_the_earlier_shape(pkg, hip, oracle, general_first):
    """pathtracer.hpp:80 on the device's kind-sorted program (axis planes are tested before general planes whatever the
    scene order): the earlier shape must win an exact tie.  The parameter of the later copy gets NO gradient at all, the
    earlier one matches the oracle; every device route agrees."""
    scene = tie_scene(pkg, general_first)
    cam = pkg.cornell_camera(48, 40)
    rp = pkg.RenderParams(spp=4, min_bounces=5, absorb=1.0, seed=9)
    ref = oracle.render(scene, cam, rp, backward=True)
    i_first, i_second = scene.n_params - 2, scene.n_params - 1
    assert np.abs(ref["grads"][i_first]).max() > 0 and np.abs(ref["grads"][i_second]).max() == 0
    hip.upload_scene(scene)
    import dataclasses
    for kw in (dict(), dict(f64=True)):
        for bpl in (0, 1):                                 # k_path (kind-sorted program) and the queue wavefront
            img, grads, stats = hip.render(cam, dataclasses.replace(rp, bounces_per_launch=bpl), backward=True, **kw)
            assert np.abs(grads[i_second]).max() == 0, (kw, bpl, grads[i_second])
            assert grad_rel_err(grads, ref["grads"]) <= (1e-9 if kw else GRAD_TOL)
            if kw:
                assert stats["segments"] == ref["stats"]["segments"]
