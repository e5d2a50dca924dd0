"""The register budgets the hot kernels are compiled for (round 6, second session: DESIGN.md section 0 item 8, profiles/r06_waves_per_simd.txt).
Each was measured with the builds alternating in one process; this test pins what the compiler makes of them -- waves per SIMD, registers,
scratch -- so that a change to a kernel header that costs a wave or brings a spill back is seen here, on the CPU box, before it is seen as
a slower frame (hipcc cross-compiles gfx950 without a GPU; ~1 minute)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CORNELL = "KindSig<24002697"          # the signature of the reference's own scene (render.cpp:39-59): the kernels the library carries


@pytest.fixture(scope="module")
def usage(tmp_path_factory):
    """{demangled kernel name: (vgprs, scratch bytes per lane, waves per SIMD, LDS bytes per block)} of the f32 / f64 path kernels"""
    subprocess.run(["python3", os.path.join(ROOT, "differentiable-renderer_amd", "csrc", "embed_sources.py")], check=True, cwd=ROOT)
    obj = str(tmp_path_factory.mktemp("res") / "drt.o")
    p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", f"-I{ROOT}/include", "-c",
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        f"{ROOT}/differentiable-renderer_amd/csrc/drt_hip.hip", "-o", obj], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?"
                      r"LDS Size \[bytes/block\]: (\d+)", p.stderr, re.S)
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"\(.*", "", d).replace("void ", ""): tuple(int(x) for x in r[1:]) for r, d in zip(rows, names)}


def one(usage, prefix, *contains):
    hits = [(k, v) for k, v in usage.items() if k.startswith(prefix) and all(c in k for c in contains)]
    assert len(hits) >= 1, (prefix, contains)
    return hits


def test_the_headline_kernel_runs_seven_waves_per_simd_without_scratch(usage):
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_path<float, false, 4, ", CORNELL, "false, false>"):
        assert (waves, scratch) == (7, 0) and vgpr <= 72, (k, vgpr, scratch, waves)
        assert lds * 7 <= 160 * 1024, (k, lds)                       # seven blocks per CU hold their LDS


def test_the_other_forms_keep_their_budgets(usage):
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_path<float, false, -1, ", CORNELL, "false, false>"):       # general form, lockstep
        assert waves == 6 and scratch == 0, (k, vgpr, scratch, waves)
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_path<float, true, 4, ", CORNELL, "false, false>"):         # with the glossy lobe
        assert waves == 6 and scratch == 0, (k, vgpr, scratch, waves)
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_path<float, false, 4, ", CORNELL, "true, false>"):         # regenerating
        assert waves == 5 and scratch == 0 and lds * 5 <= 160 * 1024, (k, vgpr, scratch, waves, lds)
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_path<double, false, 4, ", CORNELL, "false, false>"):       # f64 route
        assert waves == 4 and scratch <= 20, (k, vgpr, scratch, waves)
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_path_unbiased<float, false, 4, "):
        assert waves == 4 and scratch <= 64, (k, vgpr, scratch, waves)
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_backward<float, 0>"):                                      # K6, an albedo per face
        assert waves == 4, (k, vgpr, scratch, waves)
    for k, (vgpr, scratch, waves, lds) in one(usage, "k_backward<float, 4>"):
        assert waves == 4 and scratch <= 8, (k, vgpr, scratch, waves)


def test_the_walk_keeps_five_blocks_per_cu(usage):
    """k_intersect_mesh: 96 registers, 31 KB of LDS (a 30-entry stack per lane), 20 bytes of scratch -- and a time that hangs on the
    schedule the compiler finds for it (HISTORY.md): any other figure here wants an A/B on the GPU (tools/ab_kernel.py)."""
    (k, (vgpr, scratch, waves, lds)), = one(usage, "k_intersect_mesh<float>")
    assert (waves, vgpr) == (5, 96) and scratch <= 20 and lds * 5 <= 160 * 1024, (k, vgpr, scratch, waves, lds)
