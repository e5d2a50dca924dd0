import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle():
    o = entry.load_oracle()
    o.lib()
    return o


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["case"] = json.loads(str(d["case"]))
    return d


def case_inputs(pkg, case):
    """Rebuild (scene, camera, render params, adjoint) from the inputs a golden fixture records."""
    scene = pkg.scene_by_name(case["scene"])
    if "requires_grad" in case:
        scene.requires_grad = list(case["requires_grad"])
    cam = pkg.Camera(case["width"], case["height"], case.get("vfov", 1.3963))
    cam.look_at(case.get("eye", (0, 0, 0)), case.get("at", (0, 0, 1)))
    rp = pkg.RenderParams(spp=case["spp"], min_bounces=case["min_bounces"], absorb=case["absorb"],
                          seed=case["seed"])
    adjoint = None
    if case.get("adjoint_seed") is not None:
        adjoint = np.random.RandomState(case["adjoint_seed"]).uniform(
            -1, 2, (case["height"], case["width"], 3)).astype(np.float32)
    if case.get("target_seed") is not None:        # the target image of a per-sample squared-error loss (loss_l2)
        adjoint = np.random.RandomState(case["target_seed"]).uniform(
            0, 0.6, (case["height"], case["width"], 3)).astype(np.float32)
    return scene, cam, rp, adjoint


LOSS_GOLDENS = ["l1_loss_l2_cornell_48x32x6_d5", "l2_loss_l2_emissive_wall_40x30x5_rr", "l3_loss_l2_random3_36x28x5",
                "l4_loss_l2_mesh10x12_28x24x4", "l5_loss_l2_specular_40x32x6_rr"]


SMALL_GOLDENS = ["g2_cornell_32x32x4_d4", "g3_cornell_64x64x8_d8", "g3b_cornell_64x64x8_rr",
                 "g4_specular_64x64x8_d8", "g4b_emissive_wall_48x32x8_adj", "g7_random3_40x30x6",
                 "g8_random8_36x36x6_d5", "g9_mesh6x8_40x30x4", "g10_mesh10x12f5_32x32x4_d4",
                 "g11_mesh40x40_48x48x4_d5", "g14_mesh10x12fall_36x30x4_rr", "m1_mirror_48x48x6_d6", "m2_mirror_wall_40x32x6_rr_adj"]
# more parameters than the register form of the one-launch kernels holds (10, 12, 20, 40; vector.hpp:185-191 knows no limit)
MANY_PARAM_GOLDENS = ["p1_cornell_shapes_48x48x8_d8", "p2_params12_40x40x6_rr_adj", "p3_params20_36x36x6_d12",
                      "p4_params40_32x32x4_d6", "p6_cornell_shapes_default_roulette_40x40x8"]
SMALL_GOLDENS += MANY_PARAM_GOLDENS
# shapes of caller-defined kinds (drt_shape_kind_desc): a disc and an axis-aligned box, the harness's plugins of the reference
# ... and a BxDF of a caller-defined kind (drt_bxdf_kind_desc): a power-cosine lobe, the harness's CosLobeBxDF plugin
USER_SHAPE_GOLDENS = ["s1_disc_box_48x48x8_d6", "s2_disc_box_40x32x8_rr_adj", "b1_coslobe_48x40x8_d6", "b2_coslobe_disc_40x32x8_rr_adj"]
USER_SHAPE_UNBIASED_GOLDENS = ["s3_unbiased_disc_32x32x4_rr", "b3_unbiased_coslobe_32x28x4_rr"]


UNBIASED_GOLDENS = ["u1_unbiased_cornell_40x30x4_rr", "u2_unbiased_cornell_48x48x4_d4",
                    "u3_unbiased_specular_32x32x4_adj", "u4_unbiased_emissive_wall_32x24x4",
                    "u5_unbiased_mesh10x12_24x24x3", "u6_unbiased_mirror_32x24x4_rr"]
# ... of a 12-parameter room (the general form of the one-launch kernels).  In f32 ONE path of this frame takes another surface
# (20 of its 33,892 segments differ, on the one-launch route and the queue route alike): pinned in f64, f32 route against route
MANY_PARAM_UNBIASED_GOLDENS = ["p5_unbiased_params12_28x28x4_rr"]
# the reference's own NaN: a roulette draw of exactly 1.0 at absorb == 1
QUIRK_GOLDENS = ["q1_nan_mirror_wall_15x36x10_d5"]
# long roulette chains under the unbiased operator: the deepest trace() stands exactly at depth 64, the library's limit
DEPTH_LIMIT_GOLDENS = ["u7_unbiased_cornell_12x10x4_depth64", "u8_unbiased_specular_12x10x4_depth64",
                       "u9_unbiased_mesh10x12f5_29x28x3_depth64"]


@pytest.fixture(scope="session")
def hip(pkg):
    """One context on device 0 for the gpu tests. No fallback: raises without the HIP library."""
    r = pkg.HipRenderer(0)
    yield r
    r.close()
