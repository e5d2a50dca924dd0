"""The fp64 C restatement (oracle/) against the golden vectors captured from the UNMODIFIED
reference headers (tests/golden/, generator: oracle/gen_golden.py).  Bit-exact: the restatement
keeps the reference's operation order and shares its libm."""
import numpy as np
import pytest

from conftest import (DEPTH_LIMIT_GOLDENS, LOSS_GOLDENS, MANY_PARAM_UNBIASED_GOLDENS, QUIRK_GOLDENS, SMALL_GOLDENS, UNBIASED_GOLDENS,
                      USER_SHAPE_GOLDENS, USER_SHAPE_UNBIASED_GOLDENS,
                      case_inputs, load_golden)


@pytest.mark.parametrize("name", SMALL_GOLDENS + USER_SHAPE_GOLDENS + ["g6_libc_64x64x8_d4", "m3_mirror_libc_32x32x4_d4"])
def test_oracle_bit_exact_vs_reference_golden(pkg, oracle, name):
    g = load_golden(name)
    case = g["case"]
    scene, cam, rp, adjoint = case_inputs(pkg, case)
    libc = case.get("rng_mode", 0) == oracle.RNG_LIBC
    dump = case.get("dump_paths", 0)
    # the reference keeps tracing zero-direction rays after a light hit: needed for the libc
    # stream and to reproduce its raycast counters / vertex dumps
    r = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint,
                      rng_mode=case.get("rng_mode", 0), faithful=True, dump_paths=dump)
    np.testing.assert_array_equal(r["image"], g["image"])
    np.testing.assert_array_equal(r["grads"], g["grads"])
    assert r["stats"]["segments"] == int(g["segments"])
    assert r["stats"]["zero_dir_segments"] == int(g["zero_dir_segments"])
    if dump:
        np.testing.assert_array_equal(r["vertices"], g["vertices"])
    if not libc:
        # skipping the continuation (what the device does) changes nothing but the counters
        r2 = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint)
        np.testing.assert_array_equal(r2["image"], g["image"])
        np.testing.assert_array_equal(r2["grads"], g["grads"])
        assert r2["stats"]["segments"] == int(g["segments"]) and r2["stats"]["zero_dir_segments"] == 0


def test_per_face_colour_parameters_are_a_material_per_face(pkg, oracle):
    """drt_mesh_desc::face_param in the restatement = the same scene with a BxDF (material record) per face, bit for bit: that
    form is what the reference harness runs (fixture g14; oracle.write_scene_file spells face_param out that way)."""
    scene = pkg.scene_by_name("mesh10x12fall")
    assert scene.n_params == 4 + 216 and scene.mesh_face_param[0] is not None
    spelled = scene.with_face_params_as_materials()
    assert len(spelled.materials) == len(scene.materials) + 216 and spelled.mesh_face_param == [None]
    cam = pkg.cornell_camera(30, 22)
    for kw in (dict(min_bounces=3, absorb=1.0), dict(min_bounces=1, absorb=0.4)):
        for unbiased in (False, True):
            rp = pkg.RenderParams(spp=3, seed=8, **kw)
            a = oracle.render(scene, cam, rp, backward=True, unbiased=unbiased)
            b = oracle.render(spelled, cam, rp, backward=True, unbiased=unbiased)
            np.testing.assert_array_equal(a["image"], b["image"])
            np.testing.assert_array_equal(a["grads"], b["grads"])
            assert a["stats"]["segments"] == b["stats"]["segments"]
            assert np.count_nonzero(np.abs(a["grads"][4:]).sum(1)) > 10


@pytest.mark.parametrize("name", ["g12_gradimage_red_48x36x8_d4", "g13_gradimage_white_40x40x6_rr",
                                  "m4_mirror_gradimage_white_32x32x6"])
def test_gradient_image_bit_exact(pkg, oracle, name):
    """Per-pixel gradient of one parameter: the reference computes it by zeroing param.grad() before
    each pixel's samples (harness); the restatement must reproduce image, gradient image and totals."""
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    r = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint, grad_image_param=g["case"]["grad_image_param"])
    np.testing.assert_array_equal(r["image"], g["image"])
    np.testing.assert_array_equal(r["grad_image"], g["grad_image"])
    np.testing.assert_array_equal(r["grads"], g["grads"])
    p = g["case"]["grad_image_param"]
    np.testing.assert_allclose(r["grad_image"].sum((0, 1)) * rp.spp, r["grads"][p], rtol=1e-12)


@pytest.mark.parametrize("name", UNBIASED_GOLDENS + DEPTH_LIMIT_GOLDENS + MANY_PARAM_UNBIASED_GOLDENS + USER_SHAPE_UNBIASED_GOLDENS)
def test_unbiased_integrator_bit_exact(pkg, oracle, name):
    """integrate(..., unbiased=true): fixtures from the reference's own integration operator driven
    by the harness tracer (the reference's Pathtracer hard-codes the biased one).  Forward image
    equals the biased image; backward re-samples, so the gradients are a different sample set."""
    g = load_golden(name)
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    dump = g["case"].get("dump_paths", 0)
    r = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint, unbiased=True, dump_paths=dump)
    np.testing.assert_array_equal(r["image"], g["image"])
    np.testing.assert_array_equal(r["grads"], g["grads"])
    assert r["stats"]["segments"] == int(g["segments"])
    assert r["stats"]["zero_dir_segments"] == int(g["zero_dir_segments"])
    if dump:
        np.testing.assert_array_equal(r["vertices"], g["vertices"])
    b = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint)
    np.testing.assert_array_equal(b["image"], r["image"])
    assert not np.array_equal(b["grads"], r["grads"])


@pytest.mark.skipif("not __import__('os').path.exists('/root/reference/include/drt')")
def test_harness_tracer_is_the_reference_pathtracer(pkg, oracle):
    """The harness's own tracer (needed to reach the unbiased operator) with unbiased = false
    reproduces drt::Pathtracer bit for bit, and the zero-length-ray switch changes no value."""
    oracle.build()
    scene, cam = pkg.cornell_box(), pkg.cornell_camera(28, 20)
    rp = pkg.RenderParams(spp=4, min_bounces=2, absorb=0.4, seed=3)
    a = oracle.render_reference(scene, cam, rp, backward=True, tracer_mode=0)
    b = oracle.render_reference(scene, cam, rp, backward=True, tracer_mode=1)
    c = oracle.render_reference(scene, cam, rp, backward=True, tracer_mode=1, zero_dir_miss=True)
    for x in (b, c):
        np.testing.assert_array_equal(a["image"], x["image"])
        np.testing.assert_array_equal(a["grads"], x["grads"])
        assert a["stats"]["segments"] == x["stats"]["segments"]
    assert a["stats"]["zero_dir_segments"] == b["stats"]["zero_dir_segments"]


def test_libc_stream_known_answers(pkg, oracle):
    """numbers probed independently from the reference with its own unseeded
    rand() stream (64x64x8, -b 4 -p 1)."""
    scene = pkg.cornell_box()
    cam = pkg.cornell_camera(64, 64)
    rp = pkg.RenderParams(spp=8, min_bounces=4, absorb=1.0, seed=1)
    r = oracle.render(scene, cam, rp, backward=True, rng_mode=oracle.RNG_LIBC, faithful=True)
    n = 64 * 64 * 8
    np.testing.assert_allclose(r["image"].mean((0, 1)), [0.0486831665, 0.0464318928, 0.0438766479], rtol=2e-8)
    np.testing.assert_allclose(r["grads"][0] / n, [0.00989532471, 0.0100936127, 0.00933074951], rtol=2e-8)
    np.testing.assert_allclose(r["grads"][1] / n, [0.00573303223, 0.00527392731, 0.004947052], rtol=2e-8)
    np.testing.assert_allclose(r["grads"][2] / n, [0.0358352661, 0.0335923927, 0.0291137695], rtol=2e-8)
    np.testing.assert_allclose(r["grads"][3] / n, r["image"].mean((0, 1)), rtol=1e-12)


def test_config1_full_image(pkg, oracle):
    g = load_golden("c1_cornell_256x256x8_d4")
    scene, cam, rp, _ = case_inputs(pkg, g["case"])
    r = oracle.render(scene, cam, rp, backward=True)
    np.testing.assert_array_equal(r["image"].astype(np.float32), g["image"])
    np.testing.assert_array_equal(r["grads"], g["grads"])
    assert r["stats"]["segments"] == int(g["segments"])


def test_config3_rows_of_the_full_size_frame(pkg, oracle):
    """Config 2/3 (512x512x64, depth 8) took the reference 116 s; the restatement re-renders one
    16-row band of it (shard 5 of 32) and must reproduce the reference's row means exactly."""
    g = load_golden("c3_cornell_512x512x64_d8")
    scene, cam, rp, _ = case_inputs(pkg, g["case"])
    import dataclasses
    rps = dataclasses.replace(rp, shard=5, n_shards=32, band_rows=16)
    r = oracle.render(scene, cam, rps, backward=False)
    rows = pkg.shard_rows(512, 16, 32, 5)
    np.testing.assert_array_equal(r["image"][rows].mean(1), g["row_mean"][rows])


@pytest.mark.skipif("not __import__('os').path.exists('/root/reference/include/drt')")
def test_oracle_vs_live_reference_on_fresh_random_scenes(pkg, oracle):
    """Where the reference is present (build container), run it live on scenes that are in no
    fixture: the restatement must still be bit-identical."""
    oracle.build()
    for seed in (101, 202):
        scene = pkg.random_scene(seed)
        cam = pkg.Camera(24, 18).look_at((0, 0.2, -0.1), (0.1, 0, 1))
        rp = pkg.RenderParams(spp=3, min_bounces=1, absorb=0.3, seed=seed)
        a = oracle.render(scene, cam, rp, backward=True, faithful=True)
        b = oracle.render_reference(scene, cam, rp, backward=True)
        np.testing.assert_array_equal(a["image"], b["image"])
        np.testing.assert_array_equal(a["grads"], b["grads"])
        assert a["stats"]["segments"] == b["stats"]["segments"]


@pytest.mark.skipif("not __import__('os').path.exists('/root/reference/include/drt')")
def test_oracle_vs_live_reference_fuzz(pkg, oracle):
    """tools/fuzz_oracle.py, 150 cases: random renders -- every scene family, both integration operators, adjoint images, the
    per-sample loss, gradient images -- through the restatement and through the reference itself: bit for bit.  (Round 4 ran
    it for 60,000 cases: profiles/r04_fuzz_oracle_vs_reference.txt.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_oracle.py"), "150", "3"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0 and "FUZZ ORACLE OK: 150 renders" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("name", LOSS_GOLDENS)
def test_per_sample_squared_error_loss_bit_exact(pkg, oracle, name):
    """README.md:93-98, `loss = loss_func(radiance); loss.backward()` with loss_func = squared error against a target
    image: the fixtures come from the reference's own autograd (`auto diff = radiance - target; auto loss = diff * diff;
    loss.backward(Vec3(1))`, oracle/ref_harness.cpp `loss l2`); the restatement must reproduce them bit for bit."""
    g = load_golden(name)
    scene, cam, rp, target = case_inputs(pkg, g["case"])
    r = oracle.render(scene, cam, rp, backward=True, adjoint=target, loss_l2=True)
    assert r["stats"]["segments"] == int(g["segments"])
    np.testing.assert_array_equal(r["image"], g["image"])
    np.testing.assert_array_equal(r["grads"], g["grads"])
    # and it is a different thing from the per-pixel seed of the default mode
    lin = oracle.render(scene, cam, rp, backward=True, adjoint=target)
    assert not np.allclose(lin["grads"], r["grads"], rtol=1e-3)


@pytest.mark.parametrize("name", QUIRK_GOLDENS)
def test_the_references_own_nan_is_reproduced(pkg, oracle, name):
    """A fixture in which THE REFERENCE produced NaN: rand() == RAND_MAX makes uniform() 1.0, the roulette of an absorb == 1
    render lets the path pass, and the path is divided by its survival probability of 0 (pathtracer.hpp:128-133).  The
    restatement reproduces the fixture NaN for NaN and bit for bit elsewhere; the device deviates on purpose (it ends the path:
    tests/test_gpu_parity.py::test_absorb_one_ends_every_path_even_where_the_references_draw_is_exactly_one)."""
    g = load_golden(name)
    assert not np.isfinite(g["grads"]).all() and not np.isfinite(g["image"]).all()
    scene, cam, rp, adjoint = case_inputs(pkg, g["case"])
    r = oracle.render(scene, cam, rp, backward=True, adjoint=adjoint)
    np.testing.assert_array_equal(r["image"], g["image"])          # (NaN == NaN for assert_array_equal)
    np.testing.assert_array_equal(r["grads"], g["grads"])
    assert r["stats"]["segments"] == int(g["segments"])
