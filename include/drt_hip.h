/*
 * drt_hip.h -- C ABI of libdrt_hip.so: the MI355X (gfx950) wavefront replacement for the
 * per-pixel forward integrator + reverse-mode gradient accumulator of
 * thalesfm/differentiable-renderer.
 *
 * What each entry point replaces in the reference (all paths relative to /root/reference):
 *
 *   drt_hip_upload_scene    the scene block of src/render.cpp:26-59 (parameters, materials,
 *                           shapes, Scene vector) flattened to POD records; plugin types are
 *                           include/drt/shape.hpp:37-111 (Plane, Sphere),
 *                           include/drt/bxdf.hpp:56-124 (DiffuseBxDF, SpecularBxDF),
 *                           include/drt/emitter.hpp:15-25 (AreaEmitter).
 *   drt_hip_update_params   re-assigning the Vector<T,3,true> scene parameters
 *                           (src/render.cpp:26-29) between optimisation steps.
 *   drt_hip_render          the pixel x sample loop src/render.cpp:72-86: Camera::sample
 *                           (include/drt/camera.hpp:51-60) -> Pathtracer::trace
 *                           (include/drt/pathtracer.hpp:121-136) -> radiance.detach() /
 *                           radiance.backward(g) (include/drt/vector.hpp:256-284), with the
 *                           gradient accumulator VariableNode::backward
 *                           (include/drt/vector.hpp:185-188) as out_param_grad.
 *
 * Conventions: every function returns 0 on success or a negative drt_status; nothing throws
 * across the ABI; the caller owns every host buffer; the context owns device memory and its
 * stream; a context is not thread-safe; calls are synchronous unless DRT_RENDER_DEVICE_OUT is
 * set (then outputs are device pointers, written on the context's stream, and the call returns
 * after enqueueing unless DRT_RENDER_SYNC is also set; the path kernels of consecutive such frames
 * may run side by side on streams of the context's own -- the outputs are still written in the
 * order of the context's stream, and a device adjoint image is read in that order too).
 *
 *   drt_hip_create_group    the same path on SEVERAL GPUs of one node from one process (SURVEY 8b:
 *   drt_hip_comm_init_rank  "ctx owns device memory/streams/RCCL comms"): the rows of the frame are
 *                           dealt to the devices in interleaved bands, every device runs the whole
 *                           pipeline on its bands, and the ONE cross-device step is the reduction of
 *                           the gradient accumulator VariableNode::backward's `m_grad += grad`
 *                           (include/drt/vector.hpp:185-188): one ncclAllReduce(sum, f64, P x 3) over
 *                           xGMI, enqueued by the library on the context's stream after K7.
 *                           create_group = one process, n devices; comm_init_rank = one process per
 *                           GPU (the launcher broadcasts the 128-byte id out of band).
 *
 *   drt_hip_pin_host        nothing in the reference (its image is a `new Vector<double,3>[W*H]` the loop writes in place,
 *                           src/render.cpp:66,82): lets the device do the same -- write the frame straight into the caller's
 *                           buffer -- for callers that render frame after frame.
 *
 * There is NO CPU fallback behind this ABI: without a HIP device drt_hip_create fails with
 * DRT_ERR_NO_DEVICE.
 */
#ifndef DRT_HIP_H
#define DRT_HIP_H

#if defined(__HIPCC_RTC__)
/* compiled by hiprtc at run time (the library specialises k_path for a scene: csrc/drt_jit.h): no system headers there */
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#else
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DRT_HIP_ABI_VERSION 8

typedef enum drt_status {
    DRT_OK = 0,
    DRT_ERR_INVALID = -1,     /* bad argument / malformed scene */
    DRT_ERR_NO_DEVICE = -2,   /* no HIP device / device id out of range */
    DRT_ERR_HIP = -3,         /* a HIP runtime call failed, see drt_hip_last_error */
    DRT_ERR_NO_SCENE = -4,    /* render before upload_scene */
    DRT_ERR_OOM = -5,         /* device allocation failed */
    DRT_ERR_UNSUPPORTED = -6, /* the request is outside what the device path covers (see drt_hip_last_error) */
    DRT_ERR_COMM = -7         /* an RCCL call failed, see drt_hip_last_error */
} drt_status;

/* ---- scene description (host-side POD, doubles: the reference computes in double,
 *      src/render.cpp:22; the device converts to its compute type) ------------------------ */

enum { DRT_SHAPE_PLANE = 0, DRT_SHAPE_SPHERE = 1, DRT_SHAPE_MESH = 2,
       DRT_SHAPE_USER = 3 /* a shape of a caller-defined KIND (drt_shape_kind_desc): any analytic Shape<T> subclass,
                             shape.hpp:11-35 -- `mesh` = index into drt_scene_desc.kinds */ };
enum { DRT_BXDF_DIFFUSE = 0, DRT_BXDF_SPECULAR = 1,
       DRT_BXDF_MIRROR = 2 /* bxdf.hpp:126-144 repaired: f = 1/cos, dir = reflect(dir_in, n), pdf 1; param = -1;
                              its sample discards two draws (see drt_rng_u31) */,
       DRT_BXDF_USER = 3   /* DRT_BXDF_USER + k: a BxDF of the caller-defined kind k (drt_bxdf_kind_desc): any other subclass of
                              BxDF<T>, bxdf.hpp:12-25, whose value is colour x a scalar lobe */ };

typedef struct drt_shape_desc {
    int32_t type;      /* DRT_SHAPE_* */
    int32_t material;  /* index into materials, -1 = no BxDF (shape.hpp:26-27 returns nullptr) */
    int32_t emitter;   /* index into emitters,  -1 = no emitter (shape.hpp:29-30) */
    int32_t mesh;      /* MESH: index into meshes; otherwise ignored */
    double p[4];       /* PLANE: normal.xyz (NOT normalised, shape.hpp:58-59), offset
                          SPHERE: center.xyz, radius   MESH: unused
                          USER: values 0..3 of the shape's record (4..7: drt_scene_desc.user_params) */
} drt_shape_desc;

/* A caller-defined analytic shape (ABI v8): what a subclass of the reference's Shape<T> plugin interface (shape.hpp:11-35:
 * virtual intersect(orig, dir, double& t) and normal(point)) becomes on the device.  The two function BODIES are HIP source,
 * compiled at run time (hiprtc) into the scene's own path kernel next to the library's plane and sphere tests:
 *     template <typename R> bool intersect(const R* p, V3<R> o, V3<R> d, R& t) { <intersect_src> }
 *     template <typename R> V3<R> normal(const R* p, V3<R> P)                  { <normal_src> }
 * p = the shape's record of 8 values (drt_shape_desc.p + user_params), o / d = ray origin and (unit) direction, P = the hit
 * point; R is float or double (DRT_RENDER_F64).  Available: V3<R> with .x .y .z and + - * (component-wise, and by a scalar),
 * mk<R>(x, y, z), dot, cross, normalize, sqrt_r, rsqrt_r, div_r, abs_r, min_r, max_r, fma_r (csrc/drt_device.h).  `intersect`
 * returns whether the ray hits at a t > 0 it stores (the contract of shape.hpp:49-56); the closest hit over all shapes and the
 * first-shape-wins tie rule stay the library's (pathtracer.hpp:72-89).  At most DRT_MAX_USER_KINDS kinds per scene.  Scenes
 * with such shapes render on the one-launch path kernels (analytic scenes, any number of parameters, biased and unbiased
 * operator); a context that may not compile (drt_hip_set_specialisation(-1 / 0)) and scenes that also hold a mesh answer
 * DRT_ERR_UNSUPPORTED. */
#define DRT_MAX_USER_KINDS 2
typedef struct drt_shape_kind_desc {
    const char* name;            /* for messages (and for test checkers that know the kind by name) */
    const char* intersect_src;
    const char* normal_src;
} drt_shape_kind_desc;

/* Triangle mesh: an EXTENSION behind the Shape<T> plugin surface (the reference has no
 * triangles).  A MESH shape stands for its triangles listed in index order at
 * the shape's position in the scene, each one a Shape<T> with these semantics (pinned by the
 * brute-force `Triangle : drt::Shape<double>` of oracle/ref_harness.cpp, compiled against the
 * reference headers): two-sided Moller-Trumbore with e1 = v1 - v0, e2 = v2 - v0, hit iff t > 0
 * (the predicate of shape.hpp:55), normal = normalize(cross(e1, e2)), never flipped (like
 * Sphere, shape.hpp:105-106); on exact ties the earlier triangle wins (pathtracer.hpp:80). */
typedef struct drt_mesh_desc {
    int32_t n_vertices, n_triangles;
    const double* vertices;        /* n_vertices x 3 */
    const uint32_t* indices;       /* n_triangles x 3 */
    const int32_t* face_material;  /* n_triangles material indices, or NULL: the shape's material */
    const int32_t* face_param;     /* n_triangles colour-parameter indices, or NULL.  A face with an entry >= 0 has a BxDF of its
                                      own -- type and exponent of the face's material, colour = that parameter: the reference's
                                      `Triangle(..., std::make_shared<DiffuseBxDF<T>>(albedo_of_this_face))` per face, 50,880
                                      albedos for BASELINE config 4's mesh without 50,880 material records; -1 = the material's
                                      own colour.  (A face without material has no BxDF; a mirror has no colour.) */
} drt_mesh_desc;

typedef struct drt_material_desc {
    int32_t type;      /* DRT_BXDF_* */
    int32_t param;     /* index of the colour parameter (bxdf.hpp:82,122 m_color); MIRROR: -1 */
    double exponent;   /* SPECULAR only (bxdf.hpp:123), not differentiable.  USER: value 0 of the BxDF's record (value 1:
                          drt_scene_desc.user_bxdf_params) */
} drt_material_desc;

/* A caller-defined BxDF (ABI v8): what a subclass of the reference's BxDF<T> plugin interface (bxdf.hpp:12-25: sample(normal,
 * dir_in) -> (dir_out, pdf) and operator()(normal, dir_in, dir_out) -> f) becomes on the device, for BxDFs of the form
 * f = colour parameter x scalar (all of the reference's are).  ONE body, HIP source, compiled at run time into the scene's path kernel:
 *     template <typename R> void bxdf(const R* p, V3<R> n, V3<R> d, R u1, R u2, V3<R>& wo, R& pdf, R& bs) { <sample_src> }
 * p = the BxDF's record of 2 values, n = the surface normal as the shape returns it, d = the direction of the ARRIVING ray
 * (dir_in = -d), u1 / u2 = the two uniform draws sample() consumes, in its order (every BxDF of this build consumes exactly two:
 * drt_rng_u31); out: wo = the sampled direction, pdf = its density, bs = the scalar with f(n, -d, wo) = colour x bs.  Also
 * available: make_frame(n, t, b), sincospi_r(x, &s, &c), pow_r, sqrt_r (csrc/drt_device.h).  Same limits as caller-defined shapes. */
#define DRT_MAX_USER_BXDF_KINDS 2
typedef struct drt_bxdf_kind_desc {
    const char* name;
    const char* sample_src;
} drt_bxdf_kind_desc;

typedef struct drt_emitter_desc {
    int32_t param;     /* index of the emission parameter (emitter.hpp:24) */
    int32_t reserved;
} drt_emitter_desc;

typedef struct drt_scene_desc {
    int32_t n_shapes, n_materials, n_emitters, n_params;
    const drt_shape_desc* shapes;        /* order = Scene order: first shape wins ties,
                                            pathtracer.hpp:80 */
    const drt_material_desc* materials;
    const drt_emitter_desc* emitters;
    const double* params;                /* n_params x 3 (RGB) */
    const uint8_t* requires_grad;        /* n_params, NULL = all true */
    int32_t n_meshes;
    int32_t n_kinds;                     /* (ABI <= 7: reserved, 0) caller-defined shape kinds; -1: none, but caller-defined BxDF kinds
                                            below (n_bxdf_kinds is read); 0: none of the fields behind `meshes` is read */
    const drt_mesh_desc* meshes;
    const drt_shape_kind_desc* kinds;    /* n_kinds */
    const double* user_params;           /* n_shapes x 4: values 4..7 of every shape's record (read for USER shapes only), or NULL: zeros */
    int32_t n_bxdf_kinds, reserved2;     /* caller-defined BxDF kinds; 0: the fields below are not read (n_kinds == 0 too: neither are these) */
    const drt_bxdf_kind_desc* bxdf_kinds;
    const double* user_bxdf_params;      /* n_materials: value 1 of every material's record (read for USER BxDFs only), or NULL: zeros */
} drt_scene_desc;

typedef struct drt_camera_desc {
    int32_t width, height;
    double vfov;                          /* camera.hpp:15 default 1.3963 */
    double eye[3], forward[3], right[3], up[3];   /* as stored by Camera / look_at,
                                                     camera.hpp:29-37 */
} drt_camera_desc;

/* render flags */
#define DRT_RENDER_BACKWARD   0x1u  /* also run the tape backward pass -> out_param_grad */
#define DRT_RENDER_DEVICE_OUT 0x2u  /* out_rgb / out_param_grad / adjoint are DEVICE pointers */
#define DRT_RENDER_SYNC       0x4u  /* with DEVICE_OUT: synchronise the stream before return */
#define DRT_RENDER_TIMING     0x8u  /* bracket every kernel launch with HIP events -> stats */
#define DRT_RENDER_F64        0x10u /* compute in double on the device (verification mode) */
#define DRT_RENDER_ALLREDUCE  0x40u /* with BACKWARD, on a context that has a communicator (drt_hip_comm_init_rank):
                                       after K7 the library enqueues ONE ncclAllReduce(sum, f64, n_params x 3) on
                                       the context's stream; out_param_grad is the sum over ALL ranks' shards.
                                       Every rank of the communicator must make the call.  (A group context
                                       always reduces; the flag is implied.) */
#define DRT_RENDER_ALLREDUCE_ASYNC 0x80u /* like DRT_RENDER_ALLREDUCE, for renders with DRT_RENDER_DEVICE_OUT that follow each other
                                       without a wait in between: the all-reduce and the copy of the reduced gradient into
                                       out_param_grad are enqueued on a SECOND stream of the context and overlap the next render's
                                       kernels (the ~30-60 us of an xGMI all-reduce of 96 bytes would otherwise stand between two
                                       0.85 ms frames on every rank).  out_param_grad is valid after drt_hip_synchronize(ctx) (or
                                       with DRT_RENDER_SYNC), NOT in the order of drt_hip_stream(ctx).  Alternate between two
                                       out_param_grad buffers if every frame's gradient is wanted.  Without DRT_RENDER_DEVICE_OUT
                                       the flag means DRT_RENDER_ALLREDUCE (drt_hip_render_async overlaps its all-reduce anyway). */
#define DRT_RENDER_SERIAL     0x100u /* with DEVICE_OUT: this frame's path kernel does not overlap its neighbours' -- everything of the frame runs
                                       on the context's stream, in order.  What an optimisation loop gets, whose frame i + 1 needs the
                                       gradients of frame i (README.md:88-101); bench.py's `serial_frame`. */
#define DRT_RENDER_LOSS_L2    0x400u /* with BACKWARD: a loss that is NOT linear in the radiance, per SAMPLE -- the reference's loop
                                       `loss = loss_func(radiance); loss.backward()` (README.md:93-98) with loss_func = squared error:
                                       adjoint_rgb is read as a TARGET image (required), and every camera sample s of pixel p is
                                       back-propagated with the seed d|L_s - target_p|^2 / dL_s = 2 (L_s - target_p) instead of a
                                       per-pixel constant.  out_param_grad = d/d params of the sum over all samples of |L_s - target|^2.
                                       (The per-pixel adjoint_rgb of the default mode is exact for losses on the pixel MEAN.)
                                       Biased operator, summed gradients (not with DRT_RENDER_UNBIASED or the gradient image). */
#define DRT_RENDER_UNFUSED    0x200u /* scenes of analytic shapes: the textbook wavefront -- K1 raygen, then per bounce K2 (k_intersect) and K3
                                       (k_shade) as launches of their own over the ray queues in HBM -- instead of the fused routes
                                       (measurement / cross-check: the paths of bounces_per_launch >= 1, values equal to f32 rounding) */
#define DRT_RENDER_UNBIASED   0x20u /* with BACKWARD: the reference's unbiased integration operator
                                       (integrate.hpp:39-52, README.md:104-136): backward draws a
                                       FRESH direction at every vertex and traces a new suffix path
                                       (O(depth^2) segments).  Draw positions follow the reference
                                       with zero-length rays never hitting (oracle/ref_harness.cpp). */

typedef struct drt_render_params {
    int32_t spp;            /* samples per pixel            (args.hpp:36-43, -n) */
    int32_t min_bounces;    /* Pathtracer ctor              (args.hpp:44-51, -b) */
    double absorb;          /* Pathtracer ctor              (args.hpp:52-59, -p) */
    int32_t max_depth;      /* extension: hard cap on path vertices, 1..DRT_MAX_DEPTH; <=0 = DRT_MAX_DEPTH.
                               The reference has none (termination by roulette only): paths the cap cuts
                               short are reported in drt_hip_stats.capped_paths, so the bias is visible.
                               With absorb == 1 the cap is min_bounces itself (every path ends there). */
    uint32_t seed;          /* key of the counter RNG, see drt_rng_u31 */
    int32_t shard, n_shards, band_rows;  /* pixel sharding: image rows are cut into bands of
                               band_rows rows dealt round-robin to n_shards; this call renders
                               the bands of `shard`. n_shards <= 1 renders everything. */
    uint32_t flags;         /* DRT_RENDER_* */
    int64_t batch_paths;    /* paths in flight per wavefront batch, <=0 = default: the largest power of two whose
                             * buffers (~0.2 KB per path) fit in an eighth of the device's memory, at most 32 GB */
    int32_t bounces_per_launch; /* scenes of analytic shapes: bounces a shade launch takes a ray through in
                               registers before survivors are compacted back into the queue (1 = the
                               classic one-launch-per-bounce wavefront, HBM-bound; up to 8). <= 0 = automatic:
                               analytic scenes with at most 8 parameters take the whole path in ONE launch, in
                               registers (k_path, DRT_K_PATH: same samples, sums in another order -- equal to f32
                               rounding); otherwise as many bounces per launch as most rays are expected to survive.
                               Values >= 1 give bitwise identical results among themselves. */
    int32_t reserved;
} drt_render_params;

#define DRT_MAX_DEPTH 64

enum {
    DRT_K_RAYGEN = 0,    /* K1 */
    DRT_K_INTERSECT = 1, /* K2 */
    DRT_K_SHADE = 2,     /* K3 (+ K4 compaction fused: ballot/prefix-sum queue append) */
    DRT_K_FILM = 3,      /* K5 */
    DRT_K_BACKWARD = 4,  /* K6 */
    DRT_K_GRADREDUCE = 5,/* K7 */
    DRT_K_INTERSECT_MESH = 6, /* K2 on triangles: the BVH walk (k_intersect_mesh), timed apart from K2's analytic pass */
    DRT_K_PATH = 7,      /* K1+K2+K3+K6 in ONE launch (k_path): a path lives in registers from the eye to its end */
    DRT_K_COUNT = 8
};

typedef struct drt_hip_stats {
    uint64_t paths;                 /* camera samples traced by this call */
    uint64_t segments;              /* ray segments = raycast calls, camera ray included,
                                       zero-direction continuations excluded */
    uint64_t batches;
    double ms_total;                /* wall time of the call (host clock) */
    double ms_kernel[DRT_K_COUNT];  /* summed HIP-event time per kernel (DRT_RENDER_TIMING) */
    uint64_t launches[DRT_K_COUNT];
    uint64_t units[DRT_K_COUNT];    /* work items processed: paths (K1, K5), segments (K3, K6, k_path), rays tested by
                                       k_intersect (mesh scenes: the camera rays only), candidate rays walked by
                                       k_intersect_mesh (the rays that reach the bounds of the mesh) */
    uint64_t queue_rays_read;       /* rays the shade launches read from the queue (a fused launch keeps a ray */
    uint64_t queue_rays_written;    /* in registers over several bounces) / survivors they wrote back: 32 B each */
    uint64_t capped_paths;          /* paths still alive when they reached max_depth (cut short: 0 when the cap is
                                       the roulette's own certain kill, absorb == 1 at min_bounces); under
                                       DRT_RENDER_UNBIASED every walk counts, the camera paths' and the suffixes'.
                                       Without a user max_depth the limit is DRT_MAX_DEPTH and a path whose roulette
                                       ends it AT that depth is not cut short: its draw is consumed as the reference
                                       consumes it, and with capped_paths == 0 the render is the reference's */
    uint64_t bvh_bytes;             /* mesh scenes: bytes of the BVH (nodes + triangle records) the walk reads from */
    uint64_t path_bytes;            /* k_path launches: the bytes they write (per-range pixel sums, per-block gradient
                                       partials, per-wave counters) -- everything that kernel moves through HBM */
    uint32_t path_program;          /* k_path launches: which closest-hit program ran (DRT_PROGRAM_*; same results, different cost) */
    uint32_t reserved;
    double jit_ms;                  /* host time this context has spent compiling and loading specialised programs so far */
} drt_hip_stats;

/* k_path's closest-hit program (Pathtracer::raycast, pathtracer.hpp:72-89, over the analytic shapes) */
enum {
    DRT_PROGRAM_NONE = 0,        /* the render did not go through k_path */
    DRT_PROGRAM_SORTED = 1,      /* shape kinds read at run time: records sorted by kind in LDS, one counted loop per kind */
    DRT_PROGRAM_BUILTIN = 2,     /* kinds compiled in, the instantiation the library carries for the reference's own scene (render.cpp:39-47) */
    DRT_PROGRAM_SPECIALISED = 3  /* kinds compiled in at run time for THIS scene (hiprtc; drt_hip_set_specialisation) */
};

typedef struct drt_hip_ctx drt_hip_ctx;

int drt_hip_abi_version(void);
int drt_hip_device_count(void);
int drt_hip_create(int device_id, drt_hip_ctx** out);
/* One process, n GPUs.  The group context owns one member context per entry of device_ids
 * (device memory, stream) and the RCCL communicators between the distinct devices (ncclCommInitAll).
 * upload_scene / update_params go to every member; drt_hip_render renders member i's interleaved
 * row bands on device_ids[i] -- all members enqueued before any is waited for -- and returns the whole
 * frame in out_rgb and the gradient ALREADY summed over the members in out_param_grad: members that
 * share a device are added on that device, the distinct devices by ONE ncclAllReduce.  Host buffers only
 * (DRT_RENDER_DEVICE_OUT is refused); rp->shard / n_shards address the GROUP as one shard of a larger job
 * (multi-node), normally 0 / 1.  A device may be listed more than once (testing on a single-GPU box). */
int drt_hip_create_group(const int* device_ids, int n_devices, drt_hip_ctx** out);
int drt_hip_group_size(const drt_hip_ctx* ctx);   /* members of a group context, 1 for a plain one */
/* The PCI bus id ("0000:c1:00.0") of the device member `member` of the context renders on (0 for a plain context): what a
 * launcher logs to show that N ranks -- or the N members of a group -- really sit on N different GPUs. */
int drt_hip_device_pci_bus_id(const drt_hip_ctx* ctx, int member, char* out, int capacity);
void drt_hip_destroy(drt_hip_ctx* ctx);
/* One process per GPU: rank 0 calls drt_hip_comm_unique_id and hands the 128 bytes to the other ranks
 * out of band (torch.distributed store, MPI, a file); every rank then calls drt_hip_comm_init_rank on its
 * own context (collective: returns when all n_ranks have joined).  From then on a render with
 * DRT_RENDER_BACKWARD | DRT_RENDER_ALLREDUCE returns gradients summed over all ranks. */
#define DRT_HIP_UNIQUE_ID_BYTES 128
typedef struct drt_hip_unique_id { char bytes[DRT_HIP_UNIQUE_ID_BYTES]; } drt_hip_unique_id;   /* = ncclUniqueId */
int drt_hip_comm_unique_id(drt_hip_unique_id* out);
int drt_hip_comm_init_rank(drt_hip_ctx* ctx, const drt_hip_unique_id* id, int rank, int n_ranks);
int drt_hip_comm_size(const drt_hip_ctx* ctx);    /* ranks of the context's communicator, 0 = none */
int drt_hip_comm_destroy(drt_hip_ctx* ctx);
int drt_hip_upload_scene(drt_hip_ctx* ctx, const drt_scene_desc* scene);
int drt_hip_update_params(drt_hip_ctx* ctx, const double* params /* n_params x 3 */);
/* When a scene of analytic shapes gets a path kernel compiled for ITS shape kinds (the reference dispatches
 * Shape::intersect through a vtable per shape and ray, shape.hpp:11-35; the device wants the kinds as compile-time
 * constants: ~25 % faster than reading them at run time).  The library carries that kernel for the reference's own scene;
 * for any other it compiles one with hiprtc (~0.5 s of host time per kernel variant, cached per process), bit-identical
 * in its results to the run-time program it replaces:
 *   DRT_SPECIALISE_GENERIC the run-time program even for the reference's own scene (measurement: what specialisation buys)
 *   DRT_SPECIALISE_NEVER   never compile at run time (the reference's own scene keeps the kernel the library carries for it)
 *   DRT_SPECIALISE_AUTO    (default) once the scene has rendered 2^31 path-bounces through this context (~20 ms of
 *                          frames): small test frames never pay for a compile, a render loop does within its first frames
 *                          -- on a thread of the library's own: no frame waits for the compiler, the frames rendered
 *                          meanwhile use the run-time program (drt_hip_stats.path_program says which one ran).  The
 *                          per-sample loss (DRT_RENDER_LOSS_L2) on the one-launch route is a kernel of this kind too: its
 *                          compile starts with the first such frame, the tape route renders until it has delivered
 *   DRT_SPECIALISE_NOW     at the next render that can use it, which waits for the compile (a caller that knows it
 *                          will render many frames and wants every one of them on the fast kernel)
 * The environment variable DRT_HIP_JIT (-1 | 0 | 1 | force) sets the default of new contexts.  Group contexts: every member. */
enum { DRT_SPECIALISE_GENERIC = -1, DRT_SPECIALISE_NEVER = 0, DRT_SPECIALISE_AUTO = 1, DRT_SPECIALISE_NOW = 2 };
int drt_hip_set_specialisation(drt_hip_ctx* ctx, int mode);
/* out_rgb: width*height*3 floats, row-major, mean over spp; only the rows of this shard are
 *          written (others untouched). May be NULL.
 * adjoint_rgb: width*height*3 floats, the seed every sample of that pixel is back-propagated
 *          with; NULL = all ones (src/render.cpp:80 `radiance.backward(Vec3(1))`).
 * out_param_grad: n_params*3 doubles, SUM over this shard's samples (the caller adds it to its
 *          accumulator like vector.hpp:187 does; across shards: one sum all-reduce). */
int drt_hip_render(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                   const float* adjoint_rgb, float* out_rgb, double* out_param_grad,
                   drt_hip_stats* stats);
/* A caller that renders frame after frame into the SAME host buffers (the loop of src/render.cpp:72-90 inside an optimisation
 * loop) can hand them to the context once: drt_hip_pin_host page-locks [ptr, ptr + bytes) and maps it into the device's
 * address space (hipHostRegister), and every later render whose out_rgb (or out_grad_rgb) lies inside a pinned range gets its
 * image written THERE by the finishing kernel -- no staging block, no memcpy at the end of the call (config 3: 3 MB per
 * frame).  The range must stay allocated until drt_hip_unpin_host(ptr) or drt_hip_destroy; results are bit-identical.
 * Plain contexts (a group context's members render into their own staging blocks). */
int drt_hip_pin_host(drt_hip_ctx* ctx, void* ptr, size_t bytes);
int drt_hip_unpin_host(drt_hip_ctx* ctx, void* ptr);
#define DRT_HIP_FRAMES_IN_FLIGHT 4

/* The same call WITHOUT the wait at its end, for callers that render frame after frame (the loop of src/render.cpp:72-90
 * inside an optimisation loop): drt_hip_render_async enqueues the frame and returns a ticket; the results travel to a
 * pinned block of the context on a second stream while the NEXT frame's kernels run, and drt_hip_wait(ticket) hands them
 * to out_rgb / out_param_grad (plain memcpy) and fills `stats` (totals only; no per-kernel times).  Buffer lifetime: out_rgb,
 * out_param_grad must stay valid until drt_hip_wait returns -- they are written THERE, by the calling thread; adjoint_rgb is
 * consumed before drt_hip_render_async returns.  At most DRT_HIP_FRAMES_IN_FLIGHT (four) frames are in flight (one more
 * drt_hip_render_async before the oldest was waited for returns DRT_ERR_INVALID; with three or four in flight the path
 * kernels of consecutive frames overlap), tickets are waited for in order of issue, and drt_hip_render /
 * drt_hip_render_gradient_image refuse to run while frames are in flight.  Host buffers only; a plain (non-group) context.
 * Results are bit-identical to drt_hip_render's. */
int drt_hip_render_async(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                         const float* adjoint_rgb, float* out_rgb, double* out_param_grad, uint64_t* ticket);
int drt_hip_wait(drt_hip_ctx* ctx, uint64_t ticket, drt_hip_stats* stats /* may be NULL */);
/* Per-pixel gradient image (the figure of the reference's README.md:142-145): like drt_hip_render
 * with DRT_RENDER_BACKWARD, but instead of one summed gradient vector it returns
 *   out_grad_rgb[pixel] = mean over the pixel's samples of d(seed . radiance) / d params[param]
 * (the 3 channels of that one parameter), i.e. what `param.grad()` holds after back-propagating
 * only this pixel's samples, divided by spp.  out_rgb may be NULL. */
int drt_hip_render_gradient_image(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                                  int32_t param, const float* adjoint_rgb, float* out_rgb,
                                  float* out_grad_rgb, drt_hip_stats* stats);
/* stream the context launches on (a hipStream_t), for event timing / interop */
void* drt_hip_stream(drt_hip_ctx* ctx);
int drt_hip_synchronize(drt_hip_ctx* ctx);
const char* drt_hip_last_error(drt_hip_ctx* ctx);
const char* drt_hip_kernel_name(int k);

/* ---- the per-path counter RNG (part of the contract: the oracle, the reference harness and
 *      the device all draw from it) --------------------------------------------------------
 * Replaces drt::random::uniform (include/drt/random.hpp:7-10): the n-th rand() call made while
 * tracing camera sample `path` (= pixel*spp + sample, pixel = y*width + x) returns
 * drt_rng_u31(seed, path, n) in [0, 2^31-1]; uniform = r / 2147483647.0 (RAND_MAX).
 *
 *     draw(seed, path, n) = mix32( mix32( mix32(seed + C (path_hi + 1)) + C (n + 1) ) ^ path_lo ) >> 1
 *
 * A path's key is the 64-bit pair (stream = mix32(seed + C (path_hi + 1)), path_lo): two different paths of a render never
 * share a key -- mix32 is a bijection, so at every draw index the paths of one stream (2^32 of them: path_hi fixed) even
 * get pairwise different 32-bit values -- and no path's sequence is a shifted copy of another's: the draw index enters
 * through its own hash round h(n) = mix32(stream + C (n + 1)), the path through the XOR behind it, so "path b replays
 * path a k draws later" would need h(n) ^ h(n + k) = a ^ b for every n.  (Rounds 1-2 drew mix32(key32 + C (n + 1)) with a
 * 32-bit key per path: keys that differ by a multiple of C gave shifted copies of one sequence, and the 16.7 M paths of
 * config 3 collided outright ~3e4 times in the 2^32 key space.  tests/test_oracle_properties.py scans config 3's paths.)
 * h(n) is the same for every path of a render (the device refuses frames of more than 2^32 camera samples: path_hi = 0),
 * so a wave whose lanes stand at the same draw index computes it once, on the scalar unit. */
#if defined(__HIPCC__)
#define DRT_HD __host__ __device__ static inline
#else
#define DRT_HD static inline
#endif

DRT_HD uint32_t drt_mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

typedef struct drt_rng_key {
    uint32_t stream;   /* mix32(seed + C (path_hi + 1)): shared by all paths with the same high word */
    uint32_t lo;       /* low word of the path index */
} drt_rng_key;

DRT_HD uint32_t drt_rng_stream(uint32_t seed, uint32_t path_hi) { return drt_mix32(seed + 0x9E3779B9u * (path_hi + 1u)); }

DRT_HD drt_rng_key drt_rng_path_key(uint32_t seed, uint64_t path)
{
    drt_rng_key k;
    k.stream = drt_rng_stream(seed, (uint32_t)(path >> 32));
    k.lo = (uint32_t)path;
    return k;
}

/* h(n): the draw index's own hash round (the same for every path of a stream) */
DRT_HD uint32_t drt_rng_index_hash(uint32_t stream, uint32_t n) { return drt_mix32(stream + 0x9E3779B9u * (n + 1u)); }

/* the draw, given h(n) and the path's low word */
DRT_HD uint32_t drt_rng_combine(uint32_t index_hash, uint32_t path_lo) { return drt_mix32(index_hash ^ path_lo) >> 1; }

DRT_HD uint32_t drt_rng_draw(drt_rng_key key, uint32_t n)
{
    return drt_rng_combine(drt_rng_index_hash(key.stream, n), key.lo);
}

DRT_HD uint32_t drt_rng_u31(uint32_t seed, uint64_t path, uint32_t n)
{
    return drt_rng_draw(drt_rng_path_key(seed, path), n);
}

#ifdef __cplusplus
}
#endif
#endif /* DRT_HIP_H */
