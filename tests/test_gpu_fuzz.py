"""Random render configurations through every route of the library: the one-launch kernels (k_path and its regenerating,
8-parameter and unbiased forms) against the queue wavefront in the f64 mode -- identical segment counts, gradients to
1e-9 -- AND both against the CPU restatement of the reference, over scenes, sizes, depths, roulette settings, depth caps,
shards, batches, adjoint images, the per-sample loss and gradient images; f32 renders of the run-time and the compiled-in hit
program bit for bit (tools/fuzz_modes.py).  Random meshes against the restatement's linear scan (tools/fuzz_mesh.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_one_launch_kernels_agree_with_the_wavefront_on_random_configurations():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_modes.py"), "120", "11"], cwd=ROOT,
                         capture_output=True, text=True, timeout=550)
    assert out.returncode == 0 and "FUZZ OK: 120 cases" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.timeout(600)
def test_random_meshes_agree_with_the_restatements_linear_scan():
    """tools/fuzz_mesh.py: random displaced-sphere meshes (12 to 4,600 triangles), random frames, both operators, odd batch sizes
    -- the device's BVH walk in f64 against the restatement's scan over every triangle: identical ray counts, gradients 1e-9."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_mesh.py"), "150", "21"], cwd=ROOT,
                         capture_output=True, text=True, timeout=550)
    assert out.returncode == 0 and "FUZZ MESH OK: 150 cases" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
