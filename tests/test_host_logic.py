"""Host-side mirror of the reference's plugin vocabulary: scene flattening, camera, sharding."""
import os

import numpy as np


def test_cornell_scene_is_render_cpp(pkg):
    s = pkg.cornell_box()
    assert len(s.shapes) == 9 and len(s.materials) == 4 and len(s.emitters) == 1 and s.n_params == 4
    assert s.param_names == ["red", "green", "white", "emission"]
    assert s.params[0] == (0.5, 0.0, 0.0) and s.params[3] == (1.0, 1.0, 1.0)
    # render.cpp:39-47 order: 2 spheres, 6 planes, the light last with no BxDF
    assert [sh[0] for sh in s.shapes] == [1, 1, 0, 0, 0, 0, 0, 0, 1]
    assert s.shapes[8][1] == -1 and s.shapes[8][2] == 0
    assert s.shapes[3][3] == (1.0, 0.0, 0.1, -3.0)            # right plane, normal NOT unit
    d, keep = s.to_desc()
    assert d.n_shapes == 9 and d.shapes[8].emitter == 0 and d.materials[3].exponent == 30.0
    assert pkg.cornell_box(front_specular=True).shapes[0][1] == 3


def test_camera_look_at_matches_reference_arithmetic(pkg):
    cam = pkg.cornell_camera(640, 480)                          # render.cpp:64-65
    assert cam.forward == (0.0, 0.0, 1.0) and cam.right == (-1.0, 0.0, 0.0) and cam.up == (0.0, 1.0, 0.0)
    assert cam.vfov == 1.3963
    c2 = pkg.Camera(10, 10).look_at((1, 2, 3), (4, 6, 3))
    np.testing.assert_allclose(c2.forward, (0.6, 0.8, 0.0))
    np.testing.assert_allclose(np.dot(c2.right, c2.forward), 0, atol=1e-15)
    np.testing.assert_allclose(np.cross(c2.right, c2.forward), c2.up)


def test_shard_rows_partition(pkg):
    for (h, band, n) in [(512, 16, 8), (37, 4, 3), (5, 16, 4), (100, 1, 7)]:
        rows = [pkg.shard_rows(h, band, n, s) for s in range(n)]
        allr = np.sort(np.concatenate(rows))
        np.testing.assert_array_equal(allr, np.arange(h))
        for s in range(n):
            assert all((r // band) % n == s for r in rows[s])
    np.testing.assert_array_equal(pkg.shard_rows(9, 4, 1, 0), np.arange(9))


def test_render_params_defaults_are_the_cli_defaults(pkg):
    rp = pkg.RenderParams()
    assert (rp.spp, rp.min_bounces, rp.absorb) == (100, 1, 0.5)  # args.hpp:36-59
    d = rp.to_desc()
    assert d.spp == 100 and d.n_shards == 1 and d.flags == 0


def test_bvh_builder_and_node_encoding_known_answers(tmp_path):
    """differentiable-renderer_amd/csrc/drt_bvh.h is plain host C++: every triangle in exactly one leaf,
    inside every DECODED box on its root path (exactly and as the device's f32 fma evaluates it), the
    4-wide depth within the traversal kernel's stack bound."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "bvh_kat")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(root, "tests", "cpp", "bvh_kat.cpp"), "-o", exe],
                   check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("ok"), out


def test_integer_reduced_sincos_known_answers(tmp_path):
    """csrc/drt_sincos.h (the f32 sin/cos of phi = 2 pi u that every BxDF sample uses on the device) compiled for
    the host: a sweep of the 31-bit draw range against libm in double, max abs error < 2e-7, exact at 0 and 1/2."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sincos_kat")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-ffp-contract=off", os.path.join(root, "tests", "cpp", "sincos_kat.cpp"),
                    "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("ok"), out
