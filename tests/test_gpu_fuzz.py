"""Random render configurations through every route of the library: the one-launch kernels (k_path and its regenerating,
8-parameter and unbiased forms) against the queue wavefront in the f64 mode -- identical segment counts, gradients to
1e-9 -- over scenes, sizes, depths, roulette settings, depth caps, shards, batches and adjoint images (tools/fuzz_modes.py)."""
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
