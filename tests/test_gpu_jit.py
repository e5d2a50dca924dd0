"""Run-time specialisation of k_path (csrc/drt_jit.h): for every analytic scene the library can compile the path kernel with
the scene's shape KINDS as template constants (hiprtc), as it carries it for the reference's own scene.  The specialised
program must give the SAME BITS as the kind-sorted run-time program it replaces -- image, gradients, segment count -- so
that when a scene gets it is a matter of cost only.  Needs a real MI355X."""
import dataclasses

import numpy as np
import pytest

from test_gpu_parity import tie_scene

pytestmark = pytest.mark.gpu


def both_programs(pkg, scene, cam, rp, **kw):
    """-> ((img, grads, stats) with the run-time program, the same with the program compiled for the scene)"""
    out = []
    for mode in (pkg.SPECIALISE_GENERIC, pkg.SPECIALISE_NOW):
        r = pkg.HipRenderer(0)
        try:
            r.set_specialisation(mode)
            r.upload_scene(scene)
            out.append(r.render(cam, rp, **kw))
        finally:
            r.close()
    return out


@pytest.mark.parametrize("name", ["cornell", "cornell_specular", "cornell_mirror_wall", "cornell_emissive_wall", "random3", "random7", "random11"])
def test_specialised_program_is_bit_identical_to_the_run_time_program(pkg, name):
    scene = pkg.scene_by_name(name)
    cam = pkg.cornell_camera(96, 64)
    rp = pkg.RenderParams(spp=6, min_bounces=5, absorb=1.0, seed=3)
    (ia, ga, sa), (ib, gb, sb) = both_programs(pkg, scene, cam, rp, backward=True)
    assert sa["path_program"] == "sorted"
    # (the cornell_* variants change materials, not shape kinds: they share the signature the library carries a kernel for)
    assert sb["path_program"] == ("builtin" if name.startswith("cornell") else "specialised"), sb
    assert sa["segments"] == sb["segments"]
    assert np.array_equal(ia, ib)
    assert np.array_equal(ga, gb)


def test_specialised_program_for_a_scene_of_many_shapes(pkg):
    """more than 16 shapes: the signature's second and third word; the records no longer fit the scalar registers.
    (Also what pins the camera's constants: the run-time program takes them as the floats the HOST made of the double arguments
    (PathArgs::eye_f ..., round 6), a kernel made for more than DRT_LEAN_MAX_SHAPES shapes converts the doubles itself with
    v_cvt_f32_f64 -- the same bits, or no pixel of the two frames would agree.)"""
    scene = pkg.random_scene(4, n_spheres=36)
    assert 40 < len(scene.shapes) <= 64
    cam = pkg.cornell_camera(64, 48)
    rp = pkg.RenderParams(spp=4, min_bounces=4, absorb=1.0, seed=8)
    (ia, ga, sa), (ib, gb, sb) = both_programs(pkg, scene, cam, rp, backward=True)
    assert (sa["path_program"], sb["path_program"]) == ("sorted", "specialised")
    assert sa["segments"] == sb["segments"] and np.array_equal(ia, ib) and np.array_equal(ga, gb)


def test_specialised_roulette_and_forward_only_and_unbiased(pkg):
    """the other kernel variants: regenerating lanes (-b 1 -p 0.5), forward only, the unbiased operator"""
    scene = pkg.scene_by_name("random5")
    cam = pkg.cornell_camera(64, 64)
    for rp, kw in ((pkg.RenderParams(spp=8, min_bounces=1, absorb=0.5, seed=2), dict(backward=True)),
                   (pkg.RenderParams(spp=4, min_bounces=6, absorb=1.0, seed=2), dict(backward=False)),
                   (pkg.RenderParams(spp=2, min_bounces=4, absorb=1.0, seed=2), dict(backward=True, unbiased=True))):
        (ia, ga, sa), (ib, gb, sb) = both_programs(pkg, scene, cam, rp, **kw)
        assert (sa["path_program"], sb["path_program"]) == ("sorted", "specialised")
        assert sa["segments"] == sb["segments"]
        assert np.array_equal(ia, ib)
        if kw.get("backward"):
            assert np.array_equal(ga, gb)


@pytest.mark.parametrize("general_first", [True, False])
def test_specialised_program_keeps_the_tie_rule(pkg, oracle, general_first):
    """pathtracer.hpp:80: the earlier shape wins an exact tie -- the compiled-in program tests in scene order"""
    scene = tie_scene(pkg, general_first)
    cam = pkg.cornell_camera(48, 40)
    rp = pkg.RenderParams(spp=4, min_bounces=5, absorb=1.0, seed=9)
    ref = oracle.render(scene, cam, rp, backward=True)
    (ia, ga, sa), (ib, gb, sb) = both_programs(pkg, scene, cam, rp, backward=True)
    assert sb["path_program"] == "specialised"
    assert np.abs(gb[scene.n_params - 1]).max() == 0
    assert np.array_equal(ia, ib) and np.array_equal(ga, gb)
    assert np.abs(gb - ref["grads"]).max() <= 1e-4 * np.abs(ref["grads"]).max()


def test_automatic_specialisation_after_enough_work(pkg):
    """DRT_SPECIALISE_AUTO: small frames stay on the run-time program; once the scene has rendered 2^31 path-bounces the
    context compiles its program -- and the frames before and after agree bit for bit"""
    scene = pkg.scene_by_name("random5")
    r = pkg.HipRenderer(0)
    try:
        r.set_specialisation(pkg.SPECIALISE_AUTO)
        r.upload_scene(scene)
        small = pkg.RenderParams(spp=4, min_bounces=6, absorb=1.0, seed=5)
        cam = pkg.cornell_camera(64, 64)
        i0, g0, s0 = r.render(cam, small, backward=True)
        assert s0["path_program"] == "sorted" and s0["jit_ms"] == 0
        big = pkg.RenderParams(spp=128, min_bounces=8, absorb=1.0, seed=5)
        big_cam = pkg.cornell_camera(512, 512)
        for _ in range(9):                                   # 9 x 2^28 path-bounces
            _, _, sb = r.render(big_cam, big, backward=True)
        # the compile runs on a thread of the library's own: no frame waits for it, the frames rendered meanwhile use the
        # run-time program
        import time
        t0 = time.time()
        while True:
            i1, g1, s1 = r.render(cam, small, backward=True)
            if s1["path_program"] == "specialised" or time.time() - t0 > 60:
                break
            assert np.array_equal(i0, i1) and np.array_equal(g0, g1)
            time.sleep(0.02)
        assert s1["path_program"] == "specialised" and s1["jit_ms"] > 0
        assert np.array_equal(i0, i1) and np.array_equal(g0, g1) and s0["segments"] == s1["segments"]
    finally:
        r.close()
