"""TEST INFRASTRUCTURE ONLY: ctypes binding of the fp64 C restatement (oracle/drt_oracle.c) and a
runner for the reference harness (oracle/_ref/ref_harness, build-container only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
from typing import Optional

import numpy as np

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(ORACLE_DIR)
LIB_PATH = os.path.join(ORACLE_DIR, "libdrt_oracle.so")
REF_HARNESS = os.path.join(ORACLE_DIR, "_ref", "ref_harness")

RNG_KEYED, RNG_LIBC = 0, 1
FAITHFUL_CONTINUATION = 0x1
UNBIASED = 0x2          # integrate(..., unbiased=true): backward re-samples at every vertex
ZERO_DIR_MISS = 0x4     # zero-length rays never hit (see ref_harness.cpp)
LOSS_L2 = 0x8           # per-sample squared-error loss against a target image (README.md:93-98)


def load_pkg():
    name = "differentiable_renderer_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(REPO_ROOT, "differentiable-renderer_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class OracleStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64),
                ("zero_dir_segments", C.c_uint64), ("max_vertices", C.c_uint64), ("deepest", C.c_uint64), ("extreme_draws", C.c_uint64)]


VERTEX_DOUBLES = 16  # path, depth, o[3], d[3], shape, t, p[3], n[3]


def build(force: bool = False):
    """make libdrt_oracle.so (+ _ref/ref_harness when /root/reference is present)."""
    subprocess.run(["make", "-C", ORACLE_DIR, "libdrt_oracle.so", "ref"] + (["-B"] if force else []),
                   check=True, stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        pkg = load_pkg()
        _lib = C.CDLL(LIB_PATH)
        _lib.drt_oracle_render.argtypes = [
            C.POINTER(pkg.SceneDesc), C.POINTER(pkg.CameraDesc), C.POINTER(pkg.RenderParamsDesc),
            C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OracleStats),
            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64]
        _lib.drt_oracle_render.restype = C.c_int
        _lib.drt_oracle_set_gradient_image.argtypes = [C.c_int32, C.c_void_p]
        _lib.drt_oracle_set_gradient_image.restype = None
        _lib.drt_oracle_rng_u31.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32]
        _lib.drt_oracle_rng_u31.restype = C.c_uint32
    return _lib


def render(scene, cam, rp, backward: bool = False, adjoint: Optional[np.ndarray] = None,
           rng_mode: int = RNG_KEYED, faithful: bool = False, dump_paths: int = 0,
           grad_image_param: int = -1, unbiased: bool = False, zero_dir_miss: bool = False, loss_l2: bool = False):
    """-> dict(image f64 [H,W,3], grads f64 [P,3] | None, stats dict, vertices f64 [n,16] | None,
    grad_image f64 [H,W,3] | None)"""
    sd, keep = scene.to_desc()
    cd = cam.to_desc()
    rd = rp.to_desc()
    img = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
    grads = np.zeros((scene.n_params, 3), dtype=np.float64) if backward else None
    adj_ptr = None
    if adjoint is not None:
        adjoint = np.ascontiguousarray(adjoint, dtype=np.float32)
        adj_ptr = adjoint.ctypes.data_as(C.c_void_p)
    st = OracleStats()
    vtx = None
    nv = C.c_uint64(0)
    max_v = 0
    if dump_paths > 0:
        max_v = dump_paths * 64
        vtx = np.zeros((max_v, VERTEX_DOUBLES), dtype=np.float64)
    gimg = None
    if grad_image_param >= 0:
        assert backward
        gimg = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
        lib().drt_oracle_set_gradient_image(grad_image_param, gimg.ctypes.data_as(C.c_void_p))
    rc = lib().drt_oracle_render(C.byref(sd), C.byref(cd), C.byref(rd), rng_mode,
                                 (FAITHFUL_CONTINUATION if faithful else 0) | (UNBIASED if unbiased else 0)
                                 | (ZERO_DIR_MISS if zero_dir_miss else 0) | (LOSS_L2 if loss_l2 else 0), adj_ptr,
                                 img.ctypes.data_as(C.c_void_p),
                                 grads.ctypes.data_as(C.c_void_p) if backward else None,
                                 C.byref(st),
                                 vtx.ctypes.data_as(C.c_void_p) if vtx is not None else None,
                                 max_v, C.byref(nv), dump_paths)
    lib().drt_oracle_set_gradient_image(-1, None)
    if rc != 0:
        raise RuntimeError(f"drt_oracle_render failed: {rc}")
    return {"image": img, "grads": grads, "grad_image": gimg,
            "stats": {"paths": int(st.paths), "segments": int(st.segments),
                      "zero_dir_segments": int(st.zero_dir_segments),
                      "max_vertices": int(st.max_vertices), "deepest": int(st.deepest), "extreme_draws": int(st.extreme_draws)},
            "vertices": vtx[: nv.value] if vtx is not None else None}


def rng_u31(seed: int, path: int, n: int) -> int:
    return int(lib().drt_oracle_rng_u31(seed, path, n))


# ---- the true reference, driven by oracle/ref_harness.cpp (build container only) -------------
def have_reference() -> bool:
    return os.path.exists(REF_HARNESS)


def write_scene_file(path: str, scene, cam, rp, rng_mode: int, backward: bool, dump_paths: int,
                     adjoint_file: str = "none", grad_image_param: int = -1, tracer_mode: int = 0,
                     zero_dir_miss: bool = False, loss: str = "none"):
    if any(fp is not None for fp in getattr(scene, "mesh_face_param", [])):
        scene = scene.with_face_params_as_materials()      # (the harness knows a material per face: what the reference does)
    with open(path, "w") as f:
        f.write(f"params {len(scene.params)}\n")
        for rgb, rg in zip(scene.params, scene.requires_grad):
            f.write(f"{rgb[0]!r} {rgb[1]!r} {rgb[2]!r} {int(rg)}\n")
        f.write(f"materials {len(scene.materials)}\n")
        for i, (t, p, e) in enumerate(scene.materials):
            extra = ""
            if t >= 3:      # a BxDF of a caller-defined kind: the harness holds the same class as a plugin of the reference, by name
                extra = f" {scene.bxdf_kinds[t - 3][0]} {scene.user_m.get(i, 0.0)!r}"
            f.write(f"{t} {p} {e!r}{extra}\n")
        f.write(f"emitters {len(scene.emitters)}\n")
        for p in scene.emitters:
            f.write(f"{p}\n")
        if scene.meshes:
            f.write(f"meshes {len(scene.meshes)}\n")
            for v, idx, fm in scene.meshes:
                f.write(f"{len(v)} {len(idx)} {int(fm is not None)}\n")
                for x in v:
                    f.write(f"{float(x[0])!r} {float(x[1])!r} {float(x[2])!r}\n")
                for t in range(len(idx)):
                    f.write(f"{int(idx[t][0])} {int(idx[t][1])} {int(idx[t][2])}" + (f" {int(fm[t])}" if fm is not None else "") + "\n")
        f.write(f"shapes {len(scene.shapes)}\n")
        for i, (t, m, e, p) in enumerate(scene.shapes):
            extra = ""
            if t == 3:      # a shape of a caller-defined kind: the harness holds the same class as a plugin of the reference, by name
                kind, q = scene.user[i]
                extra = f" {scene.kinds[kind][0]} {q[0]!r} {q[1]!r} {q[2]!r} {q[3]!r}"
            f.write(f"{t} {m} {e} {p[0]!r} {p[1]!r} {p[2]!r} {p[3]!r}{extra}\n")
        v = [cam.vfov, *cam.eye, *cam.forward, *cam.right, *cam.up]
        f.write(f"camera {cam.width} {cam.height} " + " ".join(repr(float(x)) for x in v) + "\n")
        f.write(f"render {rp.spp} {rp.min_bounces} {rp.absorb!r} {rp.seed} {rng_mode} {int(backward)} {dump_paths}\n")
        f.write(f"adjoint {adjoint_file}\n")
        f.write(f"gradimage {grad_image_param}\n")
        f.write(f"mode {tracer_mode} {int(zero_dir_miss)}\n")
        f.write(f"loss {loss}\n")


def render_reference(scene, cam, rp, backward: bool = False, adjoint: Optional[np.ndarray] = None,
                     rng_mode: int = RNG_KEYED, dump_paths: int = 0, grad_image_param: int = -1,
                     tracer_mode: int = 0, zero_dir_miss: bool = False, loss_l2: bool = False):
    """tracer_mode 0 = drt::Pathtracer, 1 = the harness tracer (biased), 2 = the harness tracer with the
    reference's unbiased integration operator.
    Run the UNMODIFIED reference headers through oracle/_ref/ref_harness. Same return shape as
    render(); stats carry the harness's raycast counters and its wall time."""
    if not have_reference():
        raise RuntimeError("oracle/_ref/ref_harness not built (needs /root/reference)")
    assert rp.n_shards <= 1 and rp.max_depth <= 0, "the reference has no sharding / depth cap"
    with tempfile.TemporaryDirectory() as td:
        adj_file = "none"
        if adjoint is not None:
            adj_file = os.path.join(td, "adj.f32")
            np.ascontiguousarray(adjoint, dtype=np.float32).tofile(adj_file)
        sf = os.path.join(td, "scene.txt")
        write_scene_file(sf, scene, cam, rp, rng_mode, backward, dump_paths, adj_file, grad_image_param,
                         tracer_mode, zero_dir_miss, "l2" if loss_l2 else "none")
        prefix = os.path.join(td, "out")
        subprocess.run([REF_HARNESS, sf, prefix], check=True, stderr=subprocess.DEVNULL)
        meta = json.load(open(prefix + ".json"))
        img = np.fromfile(prefix + ".img.f64").reshape(cam.height, cam.width, 3)
        grads = np.fromfile(prefix + ".grad.f64").reshape(-1, 3) if backward else None
        vtx = None
        if dump_paths > 0:
            vtx = np.fromfile(prefix + ".vtx.f64").reshape(-1, VERTEX_DOUBLES)
        gimg = np.fromfile(prefix + ".gimg.f64").reshape(cam.height, cam.width, 3) if grad_image_param >= 0 else None
    return {"image": img, "grads": grads, "grad_image": gimg,
            "stats": {"paths": cam.width * cam.height * rp.spp,
                      "segments": meta["raycasts"] - meta["zero_dir_raycasts"],
                      "zero_dir_segments": meta["zero_dir_raycasts"], "seconds": meta["seconds"]},
            "vertices": vtx}
