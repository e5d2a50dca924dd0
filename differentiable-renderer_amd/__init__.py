"""Host-side Python mirror of the drt_hip C ABI (include/drt_hip.h).

This package is plumbing for tests and bench.py: it describes scenes with the same vocabulary
as the reference's plugin classes (Plane / Sphere, DiffuseBxDF / SpecularBxDF, AreaEmitter,
Camera.look_at, Pathtracer(absorb, min_bounces) -- /root/reference/include/drt/*.hpp), flattens
them to the POD records of the ABI and calls libdrt_hip.so through ctypes.  There is no CPU
fallback: if the HIP library or a device is missing, HipRenderer raises.

The directory name contains a hyphen, so it is loaded with importlib under the module name
``differentiable_renderer_amd`` (see __graft_entry__.load_package()).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import sys
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.path.join(PKG_DIR, "libdrt_hip.so")

# ---- enums / flags (include/drt_hip.h) ------------------------------------------------------
SHAPE_PLANE, SHAPE_SPHERE, SHAPE_MESH, SHAPE_USER = 0, 1, 2, 3
BXDF_DIFFUSE, BXDF_SPECULAR, BXDF_MIRROR, BXDF_USER = 0, 1, 2, 3
RENDER_BACKWARD = 0x1
RENDER_DEVICE_OUT = 0x2
RENDER_SYNC = 0x4
RENDER_TIMING = 0x8
RENDER_F64 = 0x10
RENDER_UNBIASED = 0x20
RENDER_SERIAL = 0x100
RENDER_UNFUSED = 0x200
RENDER_LOSS_L2 = 0x400
RENDER_ALLREDUCE = 0x40
RENDER_ALLREDUCE_ASYNC = 0x80
FRAMES_IN_FLIGHT = 4          # drt_hip_render_async: DRT_HIP_FRAMES_IN_FLIGHT
MAX_DEPTH = 64
K_RAYGEN, K_INTERSECT, K_SHADE, K_FILM, K_BACKWARD, K_GRADREDUCE, K_INTERSECT_MESH, K_PATH, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8
KERNEL_NAMES = ["raygen", "intersect", "shade", "film", "backward", "gradreduce", "intersect_mesh", "path"]
ABI_VERSION = 8
UNIQUE_ID_BYTES = 128

STATUS_NAMES = {0: "DRT_OK", -1: "DRT_ERR_INVALID", -2: "DRT_ERR_NO_DEVICE", -3: "DRT_ERR_HIP",
                -4: "DRT_ERR_NO_SCENE", -5: "DRT_ERR_OOM", -6: "DRT_ERR_UNSUPPORTED", -7: "DRT_ERR_COMM"}


class ShapeDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_int32), ("emitter", C.c_int32),
                ("mesh", C.c_int32), ("p", C.c_double * 4)]


class MeshDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_triangles", C.c_int32),
                ("vertices", C.POINTER(C.c_double)), ("indices", C.POINTER(C.c_uint32)),
                ("face_material", C.POINTER(C.c_int32)), ("face_param", C.POINTER(C.c_int32))]


class MaterialDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("param", C.c_int32), ("exponent", C.c_double)]


class EmitterDesc(C.Structure):
    _fields_ = [("param", C.c_int32), ("reserved", C.c_int32)]


class ShapeKindDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("intersect_src", C.c_char_p), ("normal_src", C.c_char_p)]


class BxdfKindDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("sample_src", C.c_char_p)]


class SceneDesc(C.Structure):
    _fields_ = [("n_shapes", C.c_int32), ("n_materials", C.c_int32), ("n_emitters", C.c_int32),
                ("n_params", C.c_int32),
                ("shapes", C.POINTER(ShapeDesc)), ("materials", C.POINTER(MaterialDesc)),
                ("emitters", C.POINTER(EmitterDesc)), ("params", C.POINTER(C.c_double)),
                ("requires_grad", C.POINTER(C.c_uint8)),
                ("n_meshes", C.c_int32), ("n_kinds", C.c_int32), ("meshes", C.POINTER(MeshDesc)),
                ("kinds", C.POINTER(ShapeKindDesc)), ("user_params", C.POINTER(C.c_double)),
                ("n_bxdf_kinds", C.c_int32), ("reserved2", C.c_int32), ("bxdf_kinds", C.POINTER(BxdfKindDesc)),
                ("user_bxdf_params", C.POINTER(C.c_double))]


class CameraDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("vfov", C.c_double),
                ("eye", C.c_double * 3), ("forward", C.c_double * 3),
                ("right", C.c_double * 3), ("up", C.c_double * 3)]


class RenderParamsDesc(C.Structure):
    _fields_ = [("spp", C.c_int32), ("min_bounces", C.c_int32), ("absorb", C.c_double),
                ("max_depth", C.c_int32), ("seed", C.c_uint32),
                ("shard", C.c_int32), ("n_shards", C.c_int32), ("band_rows", C.c_int32),
                ("flags", C.c_uint32), ("batch_paths", C.c_int64),
                ("bounces_per_launch", C.c_int32), ("reserved", C.c_int32)]


PROGRAM_NAMES = {0: "none", 1: "sorted", 2: "builtin", 3: "specialised"}   # DRT_PROGRAM_*
SPECIALISE_GENERIC, SPECIALISE_NEVER, SPECIALISE_AUTO, SPECIALISE_NOW = -1, 0, 1, 2               # drt_hip_set_specialisation


class HipStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("batches", C.c_uint64),
                ("ms_total", C.c_double), ("ms_kernel", C.c_double * K_COUNT),
                ("launches", C.c_uint64 * K_COUNT), ("units", C.c_uint64 * K_COUNT),
                ("queue_rays_read", C.c_uint64), ("queue_rays_written", C.c_uint64),
                ("capped_paths", C.c_uint64), ("bvh_bytes", C.c_uint64), ("path_bytes", C.c_uint64),
                ("path_program", C.c_uint32), ("reserved", C.c_uint32), ("jit_ms", C.c_double)]

    def as_dict(self) -> dict:
        d = {"paths": int(self.paths), "segments": int(self.segments),
             "batches": int(self.batches), "ms_total": float(self.ms_total), "kernels": {},
             "queue_rays_read": int(self.queue_rays_read), "queue_rays_written": int(self.queue_rays_written),
             "capped_paths": int(self.capped_paths), "bvh_bytes": int(self.bvh_bytes), "path_bytes": int(self.path_bytes),
             "path_program": PROGRAM_NAMES.get(int(self.path_program), str(int(self.path_program))), "jit_ms": float(self.jit_ms)}
        for k, name in enumerate(KERNEL_NAMES):
            d["kernels"][name] = {"ms": float(self.ms_kernel[k]), "launches": int(self.launches[k]),
                                  "units": int(self.units[k])}
        return d


# ---- scene vocabulary (mirrors the reference's plugin classes) ------------------------------
@dataclass
class Scene:
    """Flattened drt::Scene<T> (pathtracer.hpp:12-13) + its materials, emitters, parameters."""
    params: List[Tuple[float, float, float]] = field(default_factory=list)
    requires_grad: List[bool] = field(default_factory=list)
    param_names: List[str] = field(default_factory=list)
    materials: List[Tuple[int, int, float]] = field(default_factory=list)   # type, param, exponent
    emitters: List[int] = field(default_factory=list)                       # param
    shapes: List[Tuple[int, int, int, Tuple[float, float, float, float]]] = field(default_factory=list)
    meshes: list = field(default_factory=list)   # (vertices f64 [nv,3], indices u32 [nt,3], face_material i32 [nt] | None)
    mesh_face_param: list = field(default_factory=list)   # per mesh: colour-parameter index per face i32 [nt] (-1: the material's) | None
    # caller-defined analytic shapes (any Shape<T> subclass, shape.hpp:11-35; drt_shape_kind_desc): kinds = [(name, intersect
    # source, normal source)], user = {shape index: (kind index, values 4..7 of the shape's record)}
    kinds: list = field(default_factory=list)
    user: dict = field(default_factory=dict)
    # caller-defined BxDF kinds (any BxDF<T> subclass of the form colour x scalar lobe, bxdf.hpp:12-25; drt_bxdf_kind_desc):
    # bxdf_kinds = [(name, sample source)], user_m = {material index: value 1 of its record} (value 0 rides in `exponent`)
    bxdf_kinds: list = field(default_factory=list)
    user_m: dict = field(default_factory=dict)

    # Vector<T,3,true>(value, requires_grad), vector.hpp:228-234
    def parameter(self, rgb: Sequence[float], requires_grad: bool = True, name: str = "") -> int:
        self.params.append(tuple(float(v) for v in rgb))
        self.requires_grad.append(bool(requires_grad))
        self.param_names.append(name or f"param{len(self.params) - 1}")
        return len(self.params) - 1

    # DiffuseBxDF(color), bxdf.hpp:58-61
    def diffuse(self, param: int) -> int:
        self.materials.append((BXDF_DIFFUSE, param, 0.0))
        return len(self.materials) - 1

    # SpecularBxDF(color, exponent), bxdf.hpp:87-91
    def specular(self, param: int, exponent: float) -> int:
        self.materials.append((BXDF_SPECULAR, param, float(exponent)))
        return len(self.materials) - 1

    # AreaEmitter(emission), emitter.hpp:17
    def mirror(self) -> int:
        """drt::MirrorBxDF (bxdf.hpp:126-144, repaired): no colour parameter."""
        self.materials.append((BXDF_MIRROR, -1, 0.0))
        return len(self.materials) - 1

    def area_emitter(self, param: int) -> int:
        self.emitters.append(param)
        return len(self.emitters) - 1

    # Plane(normal, offset, bxdf, emitter), shape.hpp:40-47
    def plane(self, normal: Sequence[float], offset: float, material: int = -1, emitter: int = -1) -> int:
        self.shapes.append((SHAPE_PLANE, material, emitter,
                            (float(normal[0]), float(normal[1]), float(normal[2]), float(offset))))
        return len(self.shapes) - 1

    # Sphere(center, radius, bxdf, emitter), shape.hpp:69-76
    def sphere(self, center: Sequence[float], radius: float, material: int = -1, emitter: int = -1) -> int:
        self.shapes.append((SHAPE_SPHERE, material, emitter,
                            (float(center[0]), float(center[1]), float(center[2]), float(radius))))
        return len(self.shapes) - 1

    def bxdf_kind(self, name: str, sample_src: str) -> int:
        """A caller-defined BxDF KIND: sample + evaluate as ONE body of HIP source over a record of 2 values
        (include/drt_hip.h: drt_bxdf_kind_desc).  -> its index for user_bxdf()."""
        self.bxdf_kinds.append((name, sample_src))
        return len(self.bxdf_kinds) - 1

    def user_bxdf(self, kind: int, param: int, v0: float = 0.0, v1: float = 0.0) -> int:
        assert 0 <= kind < len(self.bxdf_kinds)
        self.materials.append((BXDF_USER + kind, param, float(v0)))
        self.user_m[len(self.materials) - 1] = float(v1)
        return len(self.materials) - 1

    def shape_kind(self, name: str, intersect_src: str, normal_src: str) -> int:
        """A caller-defined shape KIND: the bodies of Shape<T>::intersect / normal (shape.hpp:14-22) as HIP source over a record
        of 8 values (include/drt_hip.h: drt_shape_kind_desc).  -> its index for user_shape()."""
        self.kinds.append((name, intersect_src, normal_src))
        return len(self.kinds) - 1

    def user_shape(self, kind: int, record: Sequence[float], material: int = -1, emitter: int = -1) -> int:
        rec = [float(v) for v in record] + [0.0] * (8 - len(record))
        assert 0 <= kind < len(self.kinds) and len(rec) == 8
        self.shapes.append((SHAPE_USER, material, emitter, tuple(rec[:4])))
        self.user[len(self.shapes) - 1] = (kind, tuple(rec[4:]))
        return len(self.shapes) - 1

    # extension: a triangle mesh standing for its triangles at this position of the scene
    def mesh(self, vertices, indices, material: int = -1, emitter: int = -1, face_material=None, face_param=None) -> int:
        """face_param: a colour parameter of its own per face (drt_mesh_desc::face_param: the face's BxDF has the type and
        exponent of its material and this colour; -1 = the material's own)."""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 3)
        i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        fm = None if face_material is None else np.ascontiguousarray(face_material, dtype=np.int32).reshape(-1)
        fp = None if face_param is None else np.ascontiguousarray(face_param, dtype=np.int32).reshape(-1)
        assert i.max() < len(v) and (fm is None or len(fm) == len(i)) and (fp is None or len(fp) == len(i))
        self.meshes.append((v, i, fm))
        while len(self.mesh_face_param) < len(self.meshes) - 1:
            self.mesh_face_param.append(None)
        self.mesh_face_param.append(fp)
        self.shapes.append((SHAPE_MESH, material, emitter, (float(len(self.meshes) - 1), 0.0, 0.0, 0.0)))
        return len(self.shapes) - 1

    @property
    def n_params(self) -> int:
        return len(self.params)

    def with_face_params_as_materials(self) -> "Scene":
        """The same scene with every per-face colour parameter spelled out as a material of its own (what the reference
        does: a BxDF object per face) -- for the reference harness and for pinning face_param against face_material."""
        import copy
        s = copy.deepcopy(self)
        s.mesh_face_param = [None] * len(s.meshes)
        for k, (v, idx, fm) in enumerate(self.meshes):
            fp = self.mesh_face_param[k] if k < len(self.mesh_face_param) else None
            if fp is None:
                continue
            shape_mat = next(m for (t, m, e, p) in self.shapes if t == SHAPE_MESH and int(p[0]) == k)
            base = fm if fm is not None else np.full(len(idx), shape_mat, dtype=np.int32)
            new_fm = base.copy()
            made = {}
            for t in range(len(idx)):
                if fp[t] >= 0 and base[t] >= 0 and self.materials[base[t]][0] != BXDF_MIRROR:
                    key = (int(base[t]), int(fp[t]))
                    if key not in made:
                        ty, _, ex = self.materials[base[t]]
                        s.materials.append((ty, int(fp[t]), ex))
                        made[key] = len(s.materials) - 1
                    new_fm[t] = made[key]
            s.meshes[k] = (v, idx, new_fm)
        return s

    def to_desc(self):
        """-> (SceneDesc, keepalive list). Pointers stay valid while keepalive is referenced."""
        shapes = (ShapeDesc * max(1, len(self.shapes)))()
        for i, (t, m, e, p) in enumerate(self.shapes):
            shapes[i].type, shapes[i].material, shapes[i].emitter = t, m, e
            shapes[i].mesh = int(p[0]) if t == SHAPE_MESH else (self.user.get(i, (0, None))[0] if t == SHAPE_USER else 0)
            for j in range(4):
                shapes[i].p[j] = 0.0 if t == SHAPE_MESH else p[j]
        mats = (MaterialDesc * max(1, len(self.materials)))()
        for i, (t, p, ex) in enumerate(self.materials):
            mats[i].type, mats[i].param, mats[i].exponent = t, p, ex
        emis = (EmitterDesc * max(1, len(self.emitters)))()
        for i, p in enumerate(self.emitters):
            emis[i].param = p
        pa = np.ascontiguousarray(np.asarray(self.params, dtype=np.float64).reshape(-1)) if self.params else np.zeros(3)
        params = (C.c_double * max(1, 3 * len(self.params))).from_buffer_copy(pa.tobytes()) if self.params else (C.c_double * 1)()
        rga = np.asarray([1 if r else 0 for r in self.requires_grad], dtype=np.uint8)
        rg = (C.c_uint8 * max(1, len(self.params))).from_buffer_copy(rga.tobytes()) if self.params else (C.c_uint8 * 1)()
        meshes = (MeshDesc * max(1, len(self.meshes)))()
        for i, (v, idx, fm) in enumerate(self.meshes):
            meshes[i].n_vertices, meshes[i].n_triangles = len(v), len(idx)
            meshes[i].vertices = v.ctypes.data_as(C.POINTER(C.c_double))
            meshes[i].indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
            meshes[i].face_material = fm.ctypes.data_as(C.POINTER(C.c_int32)) if fm is not None else None
            fp = self.mesh_face_param[i] if i < len(self.mesh_face_param) else None
            meshes[i].face_param = fp.ctypes.data_as(C.POINTER(C.c_int32)) if fp is not None else None
        kinds = (ShapeKindDesc * max(1, len(self.kinds)))()
        for i, (name, isrc, nsrc) in enumerate(self.kinds):
            kinds[i].name, kinds[i].intersect_src, kinds[i].normal_src = name.encode(), isrc.encode(), nsrc.encode()
        uq = np.zeros((max(1, len(self.shapes)), 4), dtype=np.float64)
        for i, (_, q) in self.user.items():
            if i < len(uq):
                uq[i] = q
        bk = (BxdfKindDesc * max(1, len(self.bxdf_kinds)))()
        for i, (name, src) in enumerate(self.bxdf_kinds):
            bk[i].name, bk[i].sample_src = name.encode(), src.encode()
        um = np.zeros(max(1, len(self.materials)), dtype=np.float64)
        for i, v1 in self.user_m.items():
            if i < len(um):
                um[i] = v1
        # (n_kinds: -1 = no shape kinds, but the fields behind them -- the BxDF kinds -- are there)
        n_kinds = len(self.kinds) if self.kinds else (-1 if self.bxdf_kinds else 0)
        d = SceneDesc(len(self.shapes), len(self.materials), len(self.emitters), len(self.params),
                      shapes, mats, emis, params, rg, len(self.meshes), n_kinds, meshes,
                      kinds, uq.ctypes.data_as(C.POINTER(C.c_double)) if self.user else None,
                      len(self.bxdf_kinds), 0, bk, um.ctypes.data_as(C.POINTER(C.c_double)) if self.user_m else None)
        return d, [shapes, mats, emis, params, rg, meshes, self.meshes, self.mesh_face_param, kinds, uq, bk, um]


def cornell_box(front_specular: bool = False, emissive_wall: bool = False, front_mirror: bool = False,
                mirror_wall: bool = False, per_wall: bool = False, per_shape: bool = False) -> Scene:
    """The hard-coded scene of /root/reference/src/render.cpp:26-59 (same order, same values).

    front_specular: sphere_front uses SpecularBxDF(white, 30) (render.cpp:35 creates it, the
    reference scene leaves it unused; BASELINE config 5 uses it).
    emissive_wall: the back plane additionally carries an emitter (a shape with BOTH a BxDF and
    an emitter: several emission terms per path).
    front_mirror / mirror_wall: sphere_front / the back plane use MirrorBxDF (unit normals only: a
    mirror off the non-unit right wall would hand the spheres a non-unit direction).
    per_wall: the back, front, ground and ceiling planes get albedo parameters of their own (8 parameters in all: what
    an inverse-rendering loop over "every wall's colour" optimises).
    per_shape: EVERY shape with a BxDF gets an albedo of its own: the two spheres and the four white planes take new parameters,
    red / green keep the side walls, `white` stays declared as render.cpp:28 declares it and is no shape's colour any more --
    10 parameters, the smallest scene past the eight the register form of the one-launch kernels covers."""
    s = Scene()
    red = s.parameter((0.5, 0, 0), True, "red")                 # render.cpp:26
    green = s.parameter((0, 0.5, 0), True, "green")             # :27
    white = s.parameter((0.5, 0.5, 0.5), True, "white")         # :28
    emission = s.parameter((1, 1, 1), True, "emission")         # :29
    diffuse_red = s.diffuse(red)                                # :32
    diffuse_green = s.diffuse(green)                            # :33
    diffuse_white = s.diffuse(white)                            # :34
    specular_white = s.specular(white, 30)                      # :35
    emitter = s.area_emitter(emission)                          # :36
    mirror = s.mirror() if (front_mirror or mirror_wall) else -1
    s.sphere((0., 0., 3.), 1., mirror if front_mirror else (specular_white if front_specular else diffuse_white))  # :39
    s.sphere((-1., 1., 4.5), 1., diffuse_white)                 # :40
    s.plane((-1., 0., 0.), -3., diffuse_red)                    # :41 left
    s.plane((1., 0., 0.1), -3., diffuse_green)                  # :42 right (normal NOT unit)
    if emissive_wall:
        glow = s.parameter((0.05, 0.1, 0.2), True, "glow")
        s.plane((0., 0., -1.), -6., diffuse_white, s.area_emitter(glow))
    else:
        s.plane((0., 0., -1.), -6., mirror if mirror_wall else diffuse_white)   # :43 back
    s.plane((0, 0, 1), 0, diffuse_white)                        # :44 front
    s.plane((0., 1., 0.), -3., diffuse_white)                   # :45 ground
    s.plane((0., -1., 0.), -3., diffuse_white)                  # :46 ceiling
    if per_wall:
        for shape, (name, col) in zip((4, 5, 6, 7), (("back", (0.6, 0.5, 0.4)), ("front", (0.4, 0.4, 0.5)),
                                                     ("ground", (0.5, 0.55, 0.45)), ("ceiling", (0.45, 0.5, 0.55)))):
            t, _, e, p4 = s.shapes[shape]
            s.shapes[shape] = (t, s.diffuse(s.parameter(col, True, name)), e, p4)
    if per_shape:
        for shape, (name, col) in zip((0, 1, 4, 5, 6, 7), (("sphere_front", (0.55, 0.5, 0.45)), ("sphere_back", (0.45, 0.5, 0.6)),
                                                           ("back", (0.6, 0.5, 0.4)), ("front", (0.4, 0.4, 0.5)),
                                                           ("ground", (0.5, 0.55, 0.45)), ("ceiling", (0.45, 0.5, 0.55)))):
            t, m, e, p4 = s.shapes[shape]
            ty, _, ex = s.materials[m]
            s.materials.append((ty, s.parameter(col, True, name), ex))
            s.shapes[shape] = (t, len(s.materials) - 1, e, p4)
    s.sphere((0., 3., 3.), 1., -1, emitter)                     # :47 light (no BxDF)
    return s


# ---- two analytic shapes the library does NOT know, as a user of the reference would write them (a Shape<T> subclass each:
# oracle/ref_harness.cpp holds the same two as plugins of the unmodified reference; the CPU checker restates them) ----------
DISC_INTERSECT = """
    // record: centre p[0..2], normal p[3..5] (as given), radius p[6]
    const V3<R> c = mk<R>(p[0], p[1], p[2]), n = mk<R>(p[3], p[4], p[5]);
    const R den = dot(d, n);
    if (den == R(0)) return false;
    t = dot(c - o, n) / den;
    if (!(t > R(0))) return false;
    const V3<R> q = (o + d * t) - c;
    return dot(q, q) <= p[6] * p[6];
"""
DISC_NORMAL = """
    (void)P;
    return mk<R>(p[3], p[4], p[5]);
"""
BOX_INTERSECT = """
    // record: lower corner p[0..2], upper corner p[3..5]; slabs, nearest positive crossing
    const R lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[3], p[4], p[5]}, oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    R tn = R(-1e300), tf = R(1e300);
    if (sizeof(R) == 4) { tn = R(-3e38); tf = R(3e38); }
    for (int a = 0; a < 3; ++a) {
        const R t1 = (lo[a] - oo[a]) / dd[a], t2 = (hi[a] - oo[a]) / dd[a];
        const R ta = t1 < t2 ? t1 : t2, tb = t1 < t2 ? t2 : t1;
        tn = ta > tn ? ta : tn;
        tf = tb < tf ? tb : tf;
    }
    if (!(tn <= tf)) return false;
    t = tn > R(0) ? tn : tf;
    return t > R(0);
"""
BOX_NORMAL = """
    // the face whose plane the point lies closest to
    const R lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[3], p[4], p[5]}, pp[3] = {P.x, P.y, P.z};
    int axis = 0;
    R sign = R(-1), best = abs_r(pp[0] - lo[0]);
    for (int a = 0; a < 3; ++a) {
        const R dl = abs_r(pp[a] - lo[a]), dh = abs_r(pp[a] - hi[a]);
        if (dl < best) { best = dl; axis = a; sign = R(-1); }
        if (dh < best) { best = dh; axis = a; sign = R(1); }
    }
    return mk<R>(axis == 0 ? sign : R(0), axis == 1 ? sign : R(0), axis == 2 ? sign : R(0));
"""


COSLOBE_SAMPLE = """
    // a power-cosine lobe around the normal, record: p[0] = k.  cos(theta) = u1^(1 / (k + 1)), pdf = (k + 1) / (2 pi) cos^k,
    // f = colour (k + 2) / (2 pi) cos^k(theta_out)
    (void)d;
    const R k = p[0];
    const R ct = pow_r(u1, R(1) / (k + R(1)));
    const R st = sqrt_r(max_r(R(0), R(1) - ct * ct));
    R sphi, cphi;
    sincospi_r(R(2) * u2, &sphi, &cphi);
    V3<R> t, b;
    make_frame(n, t, b);
    wo = t * (cphi * st) + b * (sphi * st) + n * ct;
    pdf = (k + R(1)) * R(0.15915494309189535) * pow_r(ct, k);
    const R c = dot(n, wo);
    bs = c > R(0) ? (k + R(2)) * R(0.15915494309189535) * pow_r(c, k) : R(0);
"""


def cornell_with_user_bxdf(with_disc: bool = False) -> Scene:
    """The reference's scene with a material the library has NO code for on its front sphere and ground: a power-cosine lobe
    (oracle/ref_harness.cpp: CosLobeBxDF, a BxDF<T> subclass inside the unmodified reference); with_disc: a caller-defined shape too."""
    s = cornell_with_user_shapes(box=False) if with_disc else cornell_box()
    kind = s.bxdf_kind("coslobe", COSLOBE_SAMPLE)
    glossy = s.user_bxdf(kind, s.parameter((0.8, 0.7, 0.5), True, "lobe_albedo"), 6.0)
    wide = s.user_bxdf(kind, 2, 1.5)                      # (the white parameter under a wide lobe)
    t, m, e, p4 = s.shapes[0]; s.shapes[0] = (t, glossy, e, p4)      # sphere_front
    t, m, e, p4 = s.shapes[6]; s.shapes[6] = (t, wide, e, p4)        # ground
    return s


def cornell_with_user_shapes(box: bool = True) -> Scene:
    """The reference's scene (render.cpp:26-59) with two shapes of caller-defined kinds in it: a tilted diffuse disc with an
    albedo of its own, and (box) an axis-aligned box -- shapes the library has no code for."""
    s = cornell_box()
    disc = s.shape_kind("disc", DISC_INTERSECT, DISC_NORMAL)
    n = _normalize((0.2, 1.0, -0.3))
    alb = s.diffuse(s.parameter((0.7, 0.6, 0.2), True, "disc_albedo"))
    light = s.shapes.pop()                        # (the light sphere, render.cpp:47, stays last)
    s.user_shape(disc, (0.9, -1.4, 3.3, n[0], n[1], n[2], 0.9), alb)
    if box:
        bk = s.shape_kind("box", BOX_INTERSECT, BOX_NORMAL)
        s.user_shape(bk, (-2.4, -3.0, 2.2, -1.5, -1.9, 3.0), s.diffuse(s.parameter((0.3, 0.5, 0.8), True, "box_albedo")))
        s.user_shape(disc, (0.0, 2.9, 4.6, 0.0, -1.0, 0.0, 0.5), -1, s.area_emitter(s.parameter((2.0, 1.5, 1.0), True, "disc_light")))
    s.shapes.append(light)
    return s


def many_param_scene(n_params: int, n_geometry: int = 0, seed: int = 77, zero_channels: bool = True, emissive_spheres: bool = True) -> Scene:
    """A closed box of six walls and small spheres, every shape a diffuse albedo parameter: `n_params` parameters in all
    (two emissions, n_params - 2 albedos handed to the shapes round-robin).  The GEOMETRY is that of the scene with
    `n_geometry` parameters (default: n_params, at most 64) -- `many_param_scene(4, 64)` is the 64-parameter scene's room with three
    albedos and one light colour: what the parameter count alone costs.  A few albedos have zero channels (like the
    reference's red and green, render.cpp:26-27).  Test / bench input, deterministic in `seed`."""
    n_geometry = n_geometry or min(n_params, 64)
    assert 3 <= n_params <= 126 and min(n_params, 64) <= n_geometry <= 64
    rng = np.random.RandomState(seed)
    s = Scene()
    # (at most 64 materials and 63 shapes: beyond 64 parameters the extra ones are emissions of their own for the spheres)
    n_em = (2 if n_params >= 6 else 1) if n_params <= 64 else n_params - 62
    n_alb = n_params - n_em
    mats = []
    for i in range(n_alb):
        col = rng.uniform(0.25, 0.85, 3)
        if zero_channels and i % 7 == 3:
            col[rng.randint(3)] = 0.0
        mats.append(s.diffuse(s.parameter(col, True, f"albedo{i}")))
    dim = 1.0 if n_em <= 2 else 0.08
    lights = [s.area_emitter(s.parameter(rng.uniform(0.8, 3.0, 3) * (dim if i + 1 < n_em else 1.0), True, f"emission{i}")) for i in range(n_em)]
    n_bxdf_shapes = n_geometry - (2 if n_geometry >= 6 else 1)
    k = 0
    def mat():
        nonlocal k
        k += 1
        return mats[(k - 1) % n_alb]
    s.plane((-1., 0., 0.), -3., mat())
    s.plane((1., 0., 0.1), -3., mat())            # (not unit, like render.cpp:42)
    s.plane((0., 0., -1.), -6., mat())
    s.plane((0., 0., 1.), 0., mat())
    s.plane((0., 1., 0.), -3., mat())
    s.plane((0., -1., 0.), -3., mat())
    geo = np.random.RandomState(seed + 1)         # (the geometry's own stream: the same room whatever n_params)
    for i in range(max(0, n_bxdf_shapes - 6)):
        c = (geo.uniform(-2.4, 2.4), geo.uniform(-2.4, 1.6), geo.uniform(1.5, 5.4))
        s.sphere(c, geo.uniform(0.15, 0.45), mat(), lights[i] if (i < n_em - 1 and emissive_spheres) else -1)
    s.sphere((0., 3., 3.), 1., -1, lights[-1])
    return s


def random_scene(seed: int, n_spheres: int = 5, specular: bool = True, n_lights: int = 2) -> Scene:
    """A closed box with random spheres, random albedos, optional specular lobes and several
    lights (some lights also carry a BxDF). Test input only; deterministic in `seed`."""
    rng = np.random.RandomState(seed)
    s = Scene()
    mats, diffuse_mats = [], []
    for i in range(4):
        col = s.parameter(rng.uniform(0.05, 0.9, 3), bool(rng.rand() < 0.8) or i == 0, f"albedo{i}")
        diffuse_mats.append(s.diffuse(col))
        mats.append(diffuse_mats[-1])
        if specular and i % 2 == 1:
            mats.append(s.specular(col, float(rng.choice([5.0, 30.0, 80.0]))))
    lights = []
    for i in range(n_lights):
        e = s.parameter(rng.uniform(0.5, 4.0, 3), True, f"emission{i}")
        lights.append(s.area_emitter(e))
    for i in range(n_spheres):
        c = (rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(2, 5))
        s.sphere(c, rng.uniform(0.3, 0.9), mats[rng.randint(len(mats))],
                 lights[0] if (i == 0 and n_lights > 1) else -1)
    # walls: un-normalised normals on purpose (the reference never normalises them); diffuse
    # only -- the reference's specular lobe takes sqrt(1 - dot(n, h)^2) (bxdf.hpp:98), NaN for |n| > 1
    wall = lambda: diffuse_mats[rng.randint(len(diffuse_mats))]
    s.plane((-1., 0., 0.), -3., wall())
    s.plane((1., 0.05, 0.1), -3., wall())
    s.plane((0., 0., -1.), -6., wall())
    s.plane((0., 0., 1.), -0.5, wall())
    s.plane((0., 2., 0.), -6., wall())
    s.plane((0., -1., 0.), -3., wall())
    s.sphere((rng.uniform(-1, 1), 3., rng.uniform(2, 4)), 1., -1, lights[-1])
    return s


def displaced_sphere_mesh(n_lat: int, n_lon: int, center=(0., -1.2, 3.6), radius: float = 1.3,
                          amplitude: float = 0.12, seed: int = 1234):
    """Procedural test mesh (no assets, no network): a latitude/longitude sphere whose vertices are
    pushed along the radius by a smooth pseudo-random field; 2*n_lon*(n_lat-1) triangles
    (n_lat = n_lon = 160 -> 50,880: the "~50k-triangle mesh" of BASELINE config 4)."""
    rs = np.random.RandomState(seed)
    k = rs.randint(1, 6, (6, 2)).astype(np.float64)
    ph = rs.uniform(0, 2 * np.pi, (6, 2))
    amp = rs.uniform(0.3, 1.0, 6)
    verts = [(0.0, 0.0)]                                  # (theta, phi) of the north pole
    for i in range(1, n_lat):
        for j in range(n_lon):
            verts.append((np.pi * i / n_lat, 2 * np.pi * j / n_lon))
    verts.append((np.pi, 0.0))
    tp = np.array(verts)
    th, phi = tp[:, 0], tp[:, 1]
    disp = sum(a * np.sin(kk[0] * th + p[0]) * np.cos(kk[1] * phi + p[1]) for a, kk, p in zip(amp, k, ph))
    r = radius * (1.0 + amplitude * disp / amp.sum())
    xyz = np.stack([r * np.sin(th) * np.cos(phi), r * np.cos(th), r * np.sin(th) * np.sin(phi)], 1) + np.array(center)
    ring = lambda i, j: 1 + (i - 1) * n_lon + (j % n_lon)
    tris = []
    for j in range(n_lon):
        tris.append((0, ring(1, j + 1), ring(1, j)))
        for i in range(1, n_lat - 1):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            tris.append((a, b, d))
            tris.append((a, d, c))
        tris.append((len(verts) - 1, ring(n_lat - 1, j), ring(n_lat - 1, j + 1)))
    return xyz.astype(np.float64), np.array(tris, dtype=np.uint32)


def cornell_with_mesh(n_lat: int = 160, n_lon: int = 160, per_face_params: int = 0, seed: int = 1234) -> Scene:
    """BASELINE config 4 shape: the Cornell box of render.cpp with a displaced-sphere mesh in it
    (in place of sphere_front).  per_face_params > 0 gives the mesh that many albedo parameters,
    assigned to faces round-robin (per-face materials); per_face_params < 0 gives EVERY face an albedo of its own
    (drt_mesh_desc::face_param: 50,880 parameters for the 160 x 160 mesh -- config 4 as SURVEY 8d words it)."""
    s = cornell_box()
    v, idx = displaced_sphere_mesh(n_lat, n_lon, seed=seed)
    white_mat = 2
    fm = None
    fp = None
    if per_face_params > 0:
        rs = np.random.RandomState(seed + 1)
        mats = [s.diffuse(s.parameter(rs.uniform(0.2, 0.9, 3), True, f"face{i}")) for i in range(per_face_params)]
        fm = np.array([mats[t % per_face_params] for t in range(len(idx))], dtype=np.int32)
    elif per_face_params < 0:
        rs = np.random.RandomState(seed + 1)
        first = len(s.params)
        alb = rs.uniform(0.2, 0.9, (len(idx), 3))
        s.params.extend(tuple(float(x) for x in a) for a in alb)
        s.requires_grad.extend([True] * len(idx))
        s.param_names.extend(f"face{i}" for i in range(len(idx)))
        fp = np.arange(first, first + len(idx), dtype=np.int32)
    s.meshes.append((v, idx, fm))
    s.mesh_face_param.append(fp)
    s.shapes[0] = (SHAPE_MESH, white_mat, -1, (float(len(s.meshes) - 1), 0.0, 0.0, 0.0))
    return s


def scene_by_name(name: str) -> Scene:
    """Named test/bench scenes (the names the golden fixtures record)."""
    if name == "cornell":
        return cornell_box()
    if name == "cornell_specular":
        return cornell_box(front_specular=True)
    if name == "cornell_walls":
        return cornell_box(per_wall=True)
    if name == "cornell_coslobe":
        return cornell_with_user_bxdf()
    if name == "cornell_coslobe_disc":
        return cornell_with_user_bxdf(with_disc=True)
    if name == "cornell_disc":
        return cornell_with_user_shapes(box=False)
    if name == "cornell_disc_box":
        return cornell_with_user_shapes(box=True)
    if name == "cornell_shapes":
        return cornell_box(per_shape=True)
    if name.startswith("params"):        # params<n>[of<m>][x]: n parameters in the room of the m-parameter scene; x: no sphere emits
        body = name[len("params"):]
        plain = body.endswith("x")
        n, _, m = body.rstrip("x").partition("of")
        return many_param_scene(int(n), int(m) if m else 0, emissive_spheres=not plain)
    if name == "cornell_emissive_wall":
        return cornell_box(emissive_wall=True)
    if name == "cornell_mirror":
        return cornell_box(front_mirror=True)
    if name == "cornell_mirror_wall":
        return cornell_box(front_specular=True, mirror_wall=True)
    if name.startswith("random"):
        return random_scene(int(name[len("random"):]))
    if name.startswith("mesh"):          # mesh<n_lat>x<n_lon>[f<per-face params> | fall: an albedo per face]
        body, _, pf = name[4:].partition("f")
        a, b = body.split("x")
        return cornell_with_mesh(int(a), int(b), -1 if pf == "all" else (int(pf) if pf else 0))
    raise KeyError(name)


def _normalize(v):
    n = math.sqrt(((0.0 + v[0] * v[0]) + v[1] * v[1]) + v[2] * v[2])
    return (v[0] / n, v[1] / n, v[2] / n)


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


@dataclass
class Camera:
    """drt::Camera<T> (camera.hpp:10-70): same ctor defaults, same look_at arithmetic."""
    width: int
    height: int
    vfov: float = 1.3963
    eye: Tuple[float, float, float] = (0., 0., 0.)
    forward: Tuple[float, float, float] = (0., 0., -1.)
    right: Tuple[float, float, float] = (1., 0., 0.)
    up: Tuple[float, float, float] = (0., 1., 0.)

    def look_at(self, eye, at, up=(0., 1., 0.)) -> "Camera":   # camera.hpp:29-37
        self.eye = tuple(float(v) for v in eye)
        self.forward = _normalize(tuple(at[i] - eye[i] for i in range(3)))
        self.right = _normalize(_cross(self.forward, up))
        self.up = _cross(self.right, self.forward)
        return self

    def to_desc(self) -> CameraDesc:
        d = CameraDesc()
        d.width, d.height, d.vfov = self.width, self.height, self.vfov
        for i in range(3):
            d.eye[i], d.forward[i], d.right[i], d.up[i] = self.eye[i], self.forward[i], self.right[i], self.up[i]
        return d


def cornell_camera(width: int, height: int) -> Camera:
    """render.cpp:64-65."""
    return Camera(width, height).look_at((0, 0, 0), (0, 0, 1))


@dataclass
class RenderParams:
    """Pathtracer(absorb, min_bounces) (pathtracer.hpp:56-57) + the CLI's -n (args.hpp:36-59)
    + the extensions of the ABI (seed, max_depth, sharding, batching)."""
    spp: int = 100
    min_bounces: int = 1
    absorb: float = 0.5
    max_depth: int = 0
    seed: int = 1
    shard: int = 0
    n_shards: int = 1
    band_rows: int = 16
    flags: int = 0
    batch_paths: int = 0
    bounces_per_launch: int = 0     # 0 = automatic; 1 = one launch per bounce

    def to_desc(self) -> RenderParamsDesc:
        return RenderParamsDesc(self.spp, self.min_bounces, self.absorb, self.max_depth, self.seed,
                                self.shard, self.n_shards, self.band_rows, self.flags, self.batch_paths,
                                self.bounces_per_launch, 0)


def shard_rows(height: int, band_rows: int, n_shards: int, shard: int) -> np.ndarray:
    """Rows of the image owned by `shard`: bands of band_rows rows dealt round-robin."""
    y = np.arange(height)
    if n_shards <= 1:
        return y
    return y[(y // max(1, band_rows)) % n_shards == shard]


# ---- native build + loader ------------------------------------------------------------------
HIP_SOURCES = ["csrc/drt_hip.hip"]


def build_native(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> libdrt_hip.so, in-tree (cross-compiles without a GPU)."""
    srcs = [os.path.join(PKG_DIR, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(PKG_DIR, "csrc", f) for f in os.listdir(os.path.join(PKG_DIR, "csrc"))
                   if f.endswith((".h", ".hpp", ".hip"))] + [os.path.join(REPO_ROOT, "include", "drt_hip.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(REPO_ROOT, "include"), "-o", LIB_PATH] + srcs + ["-lrccl", "-lhiprtc"]
    # the device headers as strings inside the library: hiprtc specialises k_path per scene at run time (csrc/drt_jit.h)
    subprocess.run([sys.executable, os.path.join(PKG_DIR, "csrc", "embed_sources.py")], check=True)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


class DrtHipError(RuntimeError):
    pass


_ABI_SYMBOLS = ["drt_hip_abi_version", "drt_hip_device_count", "drt_hip_create", "drt_hip_create_group",
                "drt_hip_group_size", "drt_hip_device_pci_bus_id", "drt_hip_destroy",
                "drt_hip_comm_unique_id", "drt_hip_comm_init_rank", "drt_hip_comm_size", "drt_hip_comm_destroy",
                "drt_hip_upload_scene", "drt_hip_update_params", "drt_hip_set_specialisation", "drt_hip_render", "drt_hip_render_async", "drt_hip_wait",
                "drt_hip_render_gradient_image", "drt_hip_pin_host", "drt_hip_unpin_host", "drt_hip_stream",
                "drt_hip_synchronize", "drt_hip_last_error", "drt_hip_kernel_name"]


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libdrt_hip.so and declare the ABI. Raises if the library or a symbol is missing."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise DrtHipError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    for s in _ABI_SYMBOLS:
        if not hasattr(lib, s):
            raise DrtHipError(f"{path} does not export {s}")
    lib.drt_hip_abi_version.restype = C.c_int
    lib.drt_hip_device_count.restype = C.c_int
    lib.drt_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.drt_hip_create_group.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.drt_hip_group_size.argtypes = [C.c_void_p]
    lib.drt_hip_device_pci_bus_id.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    lib.drt_hip_comm_unique_id.argtypes = [C.c_void_p]
    lib.drt_hip_comm_init_rank.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.drt_hip_comm_size.argtypes = [C.c_void_p]
    lib.drt_hip_comm_destroy.argtypes = [C.c_void_p]
    lib.drt_hip_destroy.argtypes = [C.c_void_p]
    lib.drt_hip_destroy.restype = None
    lib.drt_hip_upload_scene.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    lib.drt_hip_update_params.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.drt_hip_set_specialisation.argtypes = [C.c_void_p, C.c_int]
    lib.drt_hip_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParamsDesc),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(HipStats)]
    lib.drt_hip_render_async.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParamsDesc),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.drt_hip_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(HipStats)]
    lib.drt_hip_render_gradient_image.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParamsDesc),
                                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(HipStats)]
    lib.drt_hip_pin_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.drt_hip_unpin_host.argtypes = [C.c_void_p, C.c_void_p]
    lib.drt_hip_stream.argtypes = [C.c_void_p]
    lib.drt_hip_stream.restype = C.c_void_p
    lib.drt_hip_synchronize.argtypes = [C.c_void_p]
    lib.drt_hip_last_error.argtypes = [C.c_void_p]
    lib.drt_hip_last_error.restype = C.c_char_p
    lib.drt_hip_kernel_name.argtypes = [C.c_int]
    lib.drt_hip_kernel_name.restype = C.c_char_p
    return lib


def comm_unique_id(lib_path: Optional[str] = None) -> bytes:
    """drt_hip_comm_unique_id: 128 bytes rank 0 hands to the other ranks out of band."""
    lib = load_library(lib_path)
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib.drt_hip_comm_unique_id(buf)
    if rc != 0:
        raise DrtHipError(f"drt_hip_comm_unique_id failed: {STATUS_NAMES.get(rc, rc)}")
    return buf.raw


class HipRenderer:
    """One context = one MI355X device + one stream (drt_hip_ctx); or, with a LIST of devices, a group
    context: one process driving several GPUs, gradients reduced inside the library (drt_hip_create_group)."""

    def __init__(self, device=0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        self.ctx = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*device)
            rc = self.lib.drt_hip_create_group(ids, len(device), C.byref(self.ctx))
            if rc != 0:
                raise DrtHipError(f"drt_hip_create_group(devices={list(device)}) failed: {STATUS_NAMES.get(rc, rc)}")
        else:
            rc = self.lib.drt_hip_create(device, C.byref(self.ctx))
            if rc != 0:
                raise DrtHipError(f"drt_hip_create(device={device}) failed: {STATUS_NAMES.get(rc, rc)}")
        self.device = device
        self.scene: Optional[Scene] = None
        self._pinned = []                 # arrays handed to drt_hip_pin_host (kept alive while pinned)

    @property
    def group_size(self) -> int:
        return int(self.lib.drt_hip_group_size(self.ctx))

    def pci_bus_id(self, member: int = 0) -> str:
        """the PCI bus id of the device (member `member` of) this context renders on, as the LIBRARY sees it"""
        buf = C.create_string_buffer(64)
        self._check(self.lib.drt_hip_device_pci_bus_id(self.ctx, member, buf, 64), "drt_hip_device_pci_bus_id")
        return buf.value.decode()

    def comm_init(self, unique_id: bytes, rank: int, n_ranks: int):
        """Join the communicator of a one-process-per-GPU job (collective). Afterwards renders with
        RENDER_ALLREDUCE return gradients summed over all ranks."""
        assert len(unique_id) == UNIQUE_ID_BYTES
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        self._check(self.lib.drt_hip_comm_init_rank(self.ctx, buf, rank, n_ranks), "drt_hip_comm_init_rank")

    @property
    def comm_size(self) -> int:
        return int(self.lib.drt_hip_comm_size(self.ctx))

    def comm_destroy(self):
        self._check(self.lib.drt_hip_comm_destroy(self.ctx), "drt_hip_comm_destroy")

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.drt_hip_last_error(self.ctx)
            raise DrtHipError(f"{what}: {STATUS_NAMES.get(rc, rc)}: {msg.decode() if msg else ''}")

    def upload_scene(self, scene: Scene):
        desc, keep = scene.to_desc()
        self._check(self.lib.drt_hip_upload_scene(self.ctx, C.byref(desc)), "drt_hip_upload_scene")
        self.scene = scene

    def set_specialisation(self, mode: int):
        """When the scene gets a path kernel compiled for its own shape kinds: SPECIALISE_NEVER / _AUTO / _NOW."""
        self._check(self.lib.drt_hip_set_specialisation(self.ctx, int(mode)), "drt_hip_set_specialisation")

    def update_params(self, params: np.ndarray):
        p = np.ascontiguousarray(params, dtype=np.float64).reshape(-1)
        assert self.scene is not None and p.size == 3 * self.scene.n_params
        self._check(self.lib.drt_hip_update_params(self.ctx, p.ctypes.data_as(C.POINTER(C.c_double))),
                    "drt_hip_update_params")

    def pin_host(self, array: np.ndarray):
        """drt_hip_pin_host: renders whose img_out is (inside) this array get their image written straight into it by the
        finishing kernel.  The array must outlive the pinning (unpin_host, or close())."""
        assert array.flags.c_contiguous
        self._check(self.lib.drt_hip_pin_host(self.ctx, array.ctypes.data_as(C.c_void_p), array.nbytes), "drt_hip_pin_host")
        self._pinned.append(array)

    def unpin_host(self, array: np.ndarray):
        self._check(self.lib.drt_hip_unpin_host(self.ctx, array.ctypes.data_as(C.c_void_p)), "drt_hip_unpin_host")
        self._pinned = [a for a in self._pinned if a is not array]

    def render(self, cam: Camera, rp: RenderParams, backward: bool = False,
               adjoint: Optional[np.ndarray] = None, timing: bool = False, f64: bool = False,
               unbiased: bool = False, loss_l2: bool = False, img_out: Optional[np.ndarray] = None,
               want_stats: bool = True):
        """Host-buffer render. -> (image float32 [H,W,3], grads float64 [P,3] or None, stats dict).
        loss_l2: `adjoint` is a TARGET image and every sample is back-propagated through its own squared-error loss
        (DRT_RENDER_LOSS_L2; README.md:93-98 of the reference)."""
        assert self.scene is not None
        flags = rp.flags & ~(RENDER_DEVICE_OUT | RENDER_SYNC)
        if backward:
            flags |= RENDER_BACKWARD
        if timing:
            flags |= RENDER_TIMING
        if f64:
            flags |= RENDER_F64
        if unbiased:
            flags |= RENDER_UNBIASED
        if loss_l2:
            flags |= RENDER_LOSS_L2
        d = rp.to_desc()
        d.flags = flags
        img = img_out if img_out is not None else np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        assert img.dtype == np.float32 and img.shape == (cam.height, cam.width, 3) and img.flags.c_contiguous
        grads = np.zeros((self.scene.n_params, 3), dtype=np.float64) if backward else None
        adj_ptr = None
        if adjoint is not None:
            adjoint = np.ascontiguousarray(adjoint, dtype=np.float32)
            assert adjoint.shape == (cam.height, cam.width, 3)
            adj_ptr = adjoint.ctypes.data_as(C.c_void_p)
        stats = HipStats()
        cd = cam.to_desc()
        rc = self.lib.drt_hip_render(self.ctx, C.byref(cd), C.byref(d), adj_ptr,
                                     img.ctypes.data_as(C.c_void_p),
                                     grads.ctypes.data_as(C.c_void_p) if backward else None,
                                     C.byref(stats) if (want_stats or timing) else None)
        self._check(rc, "drt_hip_render")
        return img, grads, (stats.as_dict() if (want_stats or timing) else {})

    def render_async(self, cam: Camera, rp: RenderParams, backward: bool = False, adjoint: Optional[np.ndarray] = None,
                     f64: bool = False, unbiased: bool = False, img_out: Optional[np.ndarray] = None,
                     grads_out: Optional[np.ndarray] = None):
        """drt_hip_render_async: enqueue one host-buffer frame -> a handle for wait().  At most FRAMES_IN_FLIGHT (4) frames
        in flight.  img_out / grads_out: caller-owned arrays the results are written into (a render loop keeps a set per frame in flight and spares
        itself a 3 MB allocation and its page faults per frame)."""
        assert self.scene is not None
        flags = rp.flags & ~(RENDER_DEVICE_OUT | RENDER_SYNC | RENDER_TIMING)
        flags |= (RENDER_BACKWARD if backward else 0) | (RENDER_F64 if f64 else 0) | (RENDER_UNBIASED if unbiased else 0)
        d = rp.to_desc()
        d.flags = flags
        img = img_out if img_out is not None else np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        assert img.dtype == np.float32 and img.shape == (cam.height, cam.width, 3) and img.flags.c_contiguous
        grads = None
        if backward:
            grads = grads_out if grads_out is not None else np.zeros((self.scene.n_params, 3), dtype=np.float64)
            assert grads.dtype == np.float64 and grads.shape == (self.scene.n_params, 3) and grads.flags.c_contiguous
        adj_ptr = None
        if adjoint is not None:
            adjoint = np.ascontiguousarray(adjoint, dtype=np.float32)
            assert adjoint.shape == (cam.height, cam.width, 3)
            adj_ptr = adjoint.ctypes.data_as(C.c_void_p)
        ticket = C.c_uint64(0)
        cd = cam.to_desc()
        rc = self.lib.drt_hip_render_async(self.ctx, C.byref(cd), C.byref(d), adj_ptr, img.ctypes.data_as(C.c_void_p),
                                           grads.ctypes.data_as(C.c_void_p) if backward else None, C.byref(ticket))
        self._check(rc, "drt_hip_render_async")
        return (int(ticket.value), img, grads)      # (the arrays are filled by wait())

    def wait(self, handle, want_stats: bool = True):
        """drt_hip_wait: -> (image, grads or None, stats dict) of the frame render_async() enqueued."""
        ticket, img, grads = handle
        stats = HipStats()
        self._check(self.lib.drt_hip_wait(self.ctx, ticket, C.byref(stats) if want_stats else None), "drt_hip_wait")
        return img, grads, (stats.as_dict() if want_stats else {})

    def render_gradient_image(self, cam: Camera, rp: RenderParams, param: int,
                              adjoint: Optional[np.ndarray] = None, f64: bool = False):
        """Per-pixel gradient of ONE parameter (the reference's README figure).
        -> (image float32 [H,W,3], gradient image float32 [H,W,3], stats dict)"""
        assert self.scene is not None
        d = rp.to_desc()
        d.flags = (rp.flags & ~(RENDER_DEVICE_OUT | RENDER_SYNC)) | (RENDER_F64 if f64 else 0)
        img = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        gimg = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        adj_ptr = None
        if adjoint is not None:
            adjoint = np.ascontiguousarray(adjoint, dtype=np.float32)
            adj_ptr = adjoint.ctypes.data_as(C.c_void_p)
        stats = HipStats()
        cd = cam.to_desc()
        rc = self.lib.drt_hip_render_gradient_image(self.ctx, C.byref(cd), C.byref(d), param, adj_ptr,
                                                    img.ctypes.data_as(C.c_void_p),
                                                    gimg.ctypes.data_as(C.c_void_p), C.byref(stats))
        self._check(rc, "drt_hip_render_gradient_image")
        return img, gimg, stats.as_dict()

    def render_device(self, cam: Camera, rp: RenderParams, out_rgb_ptr: int, out_grad_ptr: int,
                      adjoint_ptr: int = 0, backward: bool = True, timing: bool = False,
                      sync: bool = False, want_stats: Optional[bool] = None) -> dict:
        """Device-pointer render (outputs stay in HBM, enqueued on the context's stream).
        Without stats/timing/sync the call returns right after enqueueing (no host round trip)."""
        if want_stats is None:
            want_stats = timing
        flags = (rp.flags | RENDER_DEVICE_OUT) & ~RENDER_SYNC
        if backward:
            flags |= RENDER_BACKWARD
        if timing:
            flags |= RENDER_TIMING
        if sync:
            flags |= RENDER_SYNC
        d = rp.to_desc()
        d.flags = flags
        stats = HipStats()
        cd = cam.to_desc()
        rc = self.lib.drt_hip_render(self.ctx, C.byref(cd), C.byref(d),
                                     C.c_void_p(adjoint_ptr) if adjoint_ptr else None,
                                     C.c_void_p(out_rgb_ptr) if out_rgb_ptr else None,
                                     C.c_void_p(out_grad_ptr) if out_grad_ptr else None,
                                     C.byref(stats) if want_stats else None)
        self._check(rc, "drt_hip_render")
        return stats.as_dict() if want_stats else {}

    def synchronize(self):
        self._check(self.lib.drt_hip_synchronize(self.ctx), "drt_hip_synchronize")

    @property
    def stream(self) -> int:
        return int(self.lib.drt_hip_stream(self.ctx) or 0)

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            self.lib.drt_hip_destroy(self.ctx)
            self.ctx = C.c_void_p()
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
