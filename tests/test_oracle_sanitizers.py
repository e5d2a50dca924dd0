"""AddressSanitizer + UBSan build of the fp64 restatement, run in a child process on the small
fixtures' inputs (GPU sanitizers are unavailable on the pool; the checker at least is clean)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, os, sys
sys.path.insert(0, {root!r})
import __graft_entry__ as e
o = e.load_oracle()
o.LIB_PATH = os.path.join({root!r}, "oracle", "libdrt_oracle_asan.so")
pkg = e.load_package()
for scene in (pkg.cornell_box(), pkg.cornell_box(front_specular=True), pkg.random_scene(3), pkg.scene_by_name("mesh6x8"),
              pkg.scene_by_name("cornell_mirror_wall")):
    cam = pkg.cornell_camera(24, 18)
    rp = pkg.RenderParams(spp=3, min_bounces=2, absorb=0.3, seed=5)
    for kw in (dict(), dict(faithful=True), dict(unbiased=True), dict(grad_image_param=0), dict(dump_paths=8)):
        o.render(scene, cam, rp, backward=True, **kw)
print("sanitizers clean")
"""


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libdrt_oracle_asan.so"], check=True,
                   stdout=subprocess.DEVNULL)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "sanitizers clean" in r.stdout, r.stderr[-2000:]


def test_host_side_cxx_is_clean_under_asan_and_ubsan(tmp_path):
    """The host-side C++ that ships (BVH builder + node encoder, the drt:: headers and the flattening glue)
    through its known-answer programs, built with -fsanitize=address,undefined."""
    flags = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
             "-I" + os.path.join(ROOT, "include")]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    for src in ("bvh_kat.cpp", "host_api_kat.cpp", "dual_end_to_end.cpp"):
        exe = str(tmp_path / src.replace(".cpp", "_asan"))
        subprocess.run(flags + [os.path.join(ROOT, "tests", "cpp", src), "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, env=env)
        assert r.returncode == 0 and "ok" in r.stdout.splitlines()[-1], (src, r.stdout[-500:], r.stderr[-2000:])
