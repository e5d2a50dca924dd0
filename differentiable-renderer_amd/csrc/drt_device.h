// drt_device.h -- device-side records and math for the gfx950 wavefront path tracer.
// Written for CDNA4 only (wave64, no portability layer).
#pragma once

#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "drt_hip.h"
#if defined(__HIPCC_RTC__) && !defined(INFINITY)
#define INFINITY __builtin_huge_valf()
#endif
#include "drt_sincos.h"

#define DRT_MAX_SHAPES 64
#define DRT_MAX_MATERIALS 64
#define DRT_MAX_EMITTERS 64
#define DRT_LDS_PARAMS 256       // parameters staged in LDS by K3/K6 (more: read from L2)
#define DRT_FAST_PARAMS 8        // parameter ids accumulated in registers by K6
#ifndef DRT_PATH_LDS_PARAMS
#define DRT_PATH_LDS_PARAMS 136  // parameters the one-launch kernels stage in LDS: every analytic scene's (<= 64 materials + 64 emitters + the mirrors' constant)
#endif
#define DRT_SLOT_NONE 0xFFFFu    // DevScene::grad_slot of a parameter nobody wants a gradient for
#define DRT_ID_NONE 0xFFFFu
#define DRT_BLOCK 256
#define DRT_WAVE 64
#define DRT_PROG_SORTED_MAX DRT_MAX_SHAPES   // shapes the kind-sorted program in LDS covers (every analytic scene the ABI takes)
enum { DRT_PK_PLANE = 0, DRT_PK_SPHERE = 1, DRT_PK_AX = 2, DRT_PK_AY = 3, DRT_PK_AZ = 4,
       DRT_PK_USER0 = 5 /* ... DRT_PK_USER0 + DRT_MAX_USER_KINDS - 1: caller-defined kinds (drt_shape_kind_desc); 7 = a mesh record */ };

// ---- scene records (one instance per compute type, built by drt_hip_upload_scene) ----------
template <typename R>
struct DevShape {           // float: 32 B, one s_load_dwordx8 in K2's uniform loop
    R p[4];                 // PLANE n.xyz, offset | SPHERE c.xyz, radius
    int type, material, emitter;
    int pad;                // k_path: colour parameter | emission parameter << 16 of the shape (DRT_ID_NONE = none)
};

template <typename R>
struct DevMaterial {
    int type, param;
    R exponent;
    R norm;                 // (exponent + 2) / (2 pi), bxdf.hpp:101,118
};

template <typename R>
struct DevScene {
    int n_shapes, n_materials, n_emitters, n_params;
    unsigned long long plane_mask, sphere_mask;   // bit s: the type of shape s, tested on scalar registers
    // The same shapes as an "intersection program" for the packed f32 test (closest_hit_packed): the
    // scene order is kept, two ADJACENT planes (or spheres) share one item and are tested with
    // v_pk_* instructions, two at a time.  item i: 8 values -- a pair interleaved (x0, x1, y0, y1, z0,
    // z1, w0, w1), a single in the first four -- and one bit in each mask.
    int n_items, pad_items;
    unsigned long long item_pair, item_sphere, item_skip;   // skip: a mesh record (one shape index, no test)
    R items[DRT_MAX_SHAPES][8];
    // k_path's intersection program (drt_path.h, drt_prog.h): one record per analytic shape with a KIND.
    // prog_ok = the scene has no mesh (k_path applies).
    int prog_ok, prog_sorted;     // prog_sorted: the kind-sorted copy below is valid for the ANALYTIC shapes (mesh records left out)
    // The records SORTED BY KIND (scene order kept inside a kind): sorted[i] = (record.xyzw), sorted_shape[i] = its shape
    // index, kind k occupies [kind_begin[k], kind_begin[k + 1]).  The kind-sorted program walks them with one counted loop
    // per kind; a compiled-in program (KindSig) knows every shape's position at compile time.
    R sorted[DRT_PROG_SORTED_MAX][4];
    int sorted_shape[DRT_PROG_SORTED_MAX];
    int kind_begin[8];
    int flat[DRT_MAX_SHAPES];   // position of shape s in the flattened scene (a mesh counts once per
                                // triangle): the order that breaks exact ties, pathtracer.hpp:80
    DevShape<R> shapes[DRT_MAX_SHAPES];
    DevMaterial<R> materials[DRT_MAX_MATERIALS];
    int emitter_param[DRT_MAX_EMITTERS];
    // the one-launch kernels' gradient tables (drt_path.h, NP = DRT_NP_ANY) have a row per parameter that REQUIRES a gradient
    // (Vector<T,3,true>(value, requires_grad), vector.hpp:228-234): grad_slot[p] = its row, DRT_SLOT_NONE = none
    int n_grad_slots;
    unsigned short grad_slot[DRT_PATH_LDS_PARAMS];
    // caller-defined shapes (DRT_SHAPE_USER; DevShape::type = DRT_SHAPE_USER + kind): values 4..7 of their records
    R user_q[DRT_MAX_SHAPES][4];
};

// ---- 16-byte (f32) / 32-byte (f64) queue lanes ----------------------------------------------
template <typename R> struct Q4;
template <> struct Q4<float> { typedef float4 T; };
template <> struct Q4<double> { typedef double4 T; };

// second and third queue lanes: (d.y, d.z) is all K2 still needs after ray_a, the ids ride apart
template <typename R> struct Q2;
template <> struct Q2<float> { typedef float2 T; };
template <> struct Q2<double> { typedef double2 T; };

template <typename R> struct HitRec;
template <> struct __attribute__((aligned(8))) HitRec<float> { float t; int prim; };
template <> struct __attribute__((aligned(16))) HitRec<double> { double t; int prim; int pad; };

// one tape record per path vertex (backward only): T_{k+1} = T_k * color(ids & 0xFFFF) * m
template <typename R> struct TapeRec;
template <> struct __attribute__((aligned(8))) TapeRec<float> { float m; uint32_t ids; };
template <> struct __attribute__((aligned(16))) TapeRec<double> { double m; uint32_t ids; uint32_t pad; };

__device__ inline float pid_pack(float, uint32_t pid) { return __uint_as_float(pid); }
__device__ inline double pid_pack(double, uint32_t pid) { return (double)pid; }
__device__ inline uint32_t pid_unpack(float v) { return __float_as_uint(v); }
__device__ inline uint32_t pid_unpack(double v) { return (uint32_t)v; }

// ---- triangle meshes (extension): one BVH over all triangles of the scene ---------------------
// 4-wide node i = four 16-byte words (64 B whatever the compute type; drt_bvh.h: QNode): child boxes
// are 8-bit offsets on a per-node power-of-two grid, decoded with one fma per bound
//   node[i][0] = (origin.xyz as f32 bits, grid exponent bytes ex | ey << 8 | ez << 16)
//   node[i][1] = links of children 0..3     bit 31 clear = interior node index,
//                                           bit 31 set   = leaf: first << 3 | count (0 = no child)
//   node[i][2] = (q_lo.x, q_lo.y, q_lo.z, q_hi.x)   one byte per child in each word
//   node[i][3] = (q_hi.y, q_hi.z, -, -)
//   tri[j][0..2] (leaf order) = (v0.xyz, e1.x) (e1.yz, e2.xy) (e2.z, global index, flat index, -): ONE 48-byte record per
//                triangle -- as three arrays a leaf visit touched three cache lines, as records one or two, and a line
//                is what a visit pays for (walk 4.17 -> 4.01 ms, profiles/r03_walk_experiments.txt)
//   tri_shade[g] (global triangle order) = (normal.xyz, colour parameter | material << 16 | emitter << 24): the face's own
//                colour parameter (drt_mesh_desc::face_param) or its material's; 0xFFFF / 0xFF = none
template <typename R>
struct DevBvh {
    const uint4* node;                  // [n_nodes][4]
    const typename Q4<R>::T* tri;       // [n_tris][3]
    const typename Q4<R>::T* tri_shade;
    uint32_t n_nodes, n_top, n_tris, pad;
    R lo[3], hi[3];                     // bounds of all triangles (padded like the node boxes): K2's analytic pass uses
                                        // them to hand only the rays that can reach a triangle to the BVH walk
};
#define DRT_BVH_NONE 0xFFFFFFFFu
#define DRT_BVH_LEAF 0x80000000u
#ifndef DRT_BVH_LDS_NODES
#define DRT_BVH_LDS_NODES 16         // 1 KB of LDS per block: the very top of the tree (more measured no faster: HISTORY.md 3c)
#endif
#ifndef DRT_BVH_STACK
#define DRT_BVH_STACK 30             // per-lane traversal stack in LDS, 30 KB per block (+ 1 KB of nodes: five blocks per CU).  The
                                     // builder measures what a tree needs and rebuilds with a tighter depth bound until it fits
                                     // (drt_bvh.h; the 50,880-triangle test mesh needs 29)
#endif
#ifndef DRT_WALK_MIN_BLOCKS
#define DRT_WALK_MIN_BLOCKS 5        // k_intersect_mesh<float> is compiled for five blocks per CU (96 registers: HISTORY.md 3c)
#endif
// the walk's waves pull candidate lists from DRT_PULL_COUNTERS counters, each on a cache line of its own (one
// address sustains only ~88 returning atomics per microsecond): counter c hands out the lists c, c + 64, c + 128, ...
#define DRT_PULL_COUNTERS 64
#define DRT_PULL_STRIDE 32           // words between counters (128 bytes)
#define DRT_PULL_WORDS (DRT_PULL_STRIDE * (DRT_PULL_COUNTERS + 1))   // after the n_lists list lengths (+ alignment slack)
#define DRT_BVH_REFILL 16            // idle lanes needed before the wave pulls new rays from its stream
#ifndef DRT_BVH_DESCEND_MIN
#define DRT_BVH_DESCEND_MIN 32       // the interior-node loop runs while at least this many lanes descend (sweep: 28..40 flat)
#endif

typedef float drt_f2 __attribute__((ext_vector_type(2)));
typedef float drt_f3 __attribute__((ext_vector_type(3)));
typedef drt_f3 drt_f3_u __attribute__((aligned(4)));      // a 12-byte pixel at any float boundary (global_store_dwordx3)

// read-once streams: non-temporal loads (they go through the caches without claiming room in them)
typedef float drt_f4 __attribute__((ext_vector_type(4)));
typedef double drt_d2 __attribute__((ext_vector_type(2)));
typedef double drt_d4 __attribute__((ext_vector_type(4)));
__device__ inline float4 nt_load(const float4* p)
{
    const drt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const drt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline float2 nt_load(const float2* p)
{
    const drt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const drt_f2*>(p));
    return make_float2(v.x, v.y);
}
__device__ inline double4 nt_load(const double4* p)
{
    const drt_d4 v = __builtin_nontemporal_load(reinterpret_cast<const drt_d4*>(p));
    return make_double4(v.x, v.y, v.z, v.w);
}
__device__ inline double2 nt_load(const double2* p)
{
    const drt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const drt_d2*>(p));
    return make_double2(v.x, v.y);
}

#define DRT_PI 3.14159265358979323846
#define DRT_RAND_MAX_D 2147483647.0

// ---- small vector math -----------------------------------------------------------------------
template <typename R>
struct V3 {
    R x, y, z;
};
template <typename R> __device__ inline V3<R> mk(R x, R y, R z) { V3<R> v = {x, y, z}; return v; }
template <typename R> __device__ inline V3<R> operator+(V3<R> a, V3<R> b) { return mk<R>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename R> __device__ inline V3<R> operator-(V3<R> a, V3<R> b) { return mk<R>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename R> __device__ inline V3<R> operator*(V3<R> a, V3<R> b) { return mk<R>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename R> __device__ inline V3<R> operator*(V3<R> a, R s) { return mk<R>(a.x * s, a.y * s, a.z * s); }
template <typename R> __device__ inline V3<R> operator-(V3<R> a) { return mk<R>(-a.x, -a.y, -a.z); }
template <typename R> __device__ inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename R> __device__ inline V3<R> cross(V3<R> a, V3<R> b)
{
    return mk<R>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// f32: the hardware's 1-ulp v_rsq / v_sqrt / v_rcp (no IEEE refinement sequences: they are
// ~10 VALU instructions each and the path is already tolerance-, not bit-, comparable to the
// fp64 reference).  f64 (verification mode): correctly rounded.
__device__ inline float rsqrt_r(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ inline float sqrt_r(float x) { return __builtin_amdgcn_sqrtf(x); }
// f64: the library's own refinement of v_rsq_f64 (Goldschmidt step + two corrections: what clang emits for sqrt()) WITHOUT its
// range scaling (ldexp in, ldexp out, class test: 7 of its 17 instructions) -- the arguments here are squared lengths and
// discriminants of a scene a few units across, nowhere near 2^-767.  Same result for every such argument.
#ifndef DRT_F64_LIBM
__device__ inline double sqrt_r(double x)
{
    if (!(x > 0.0))
        return x == 0.0 ? 0.0 : __builtin_nan("");
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d0 = fma(-g, g, x);
    g = fma(d0, h, g);
    const double d1 = fma(-g, g, x);
    return fma(d1, h, g);
}
// 1 / sqrt(x) to the last bit or two (the reference forms 1.0 / sqrt(x): a correctly rounded root, then a correctly rounded
// quotient -- 29 instructions; this is v_rsq_f64 and two Newton steps in the residual form, 8): the f64 mode's bound is 1e-9
__device__ inline double rsqrt_r(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);                     // 1 - x y^2
    y = fma(y * e, fma(e, 0.375, 0.5), y);              // y (1 + e/2 + 3 e^2 / 8)
    e = fma(-x * y, y, 1.0);
    return fma(y * e, 0.5, y);
}
#else
__device__ inline double sqrt_r(double x) { return sqrt(x); }
__device__ inline double rsqrt_r(double x) { return 1.0 / sqrt(x); }
#endif
__device__ inline float div_r(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ inline double div_r(double a, double b) { return a / b; }
__device__ inline float min_r(float a, float b) { return fminf(a, b); }   // NaN-ignoring (v_min_f32)
__device__ inline double min_r(double a, double b) { return fmin(a, b); }
__device__ inline float fma_r(float a, float b, float c) { return fmaf(a, b, c); }
__device__ inline double fma_r(double a, double b, double c) { return fma(a, b, c); }
__device__ inline float max_r(float a, float b) { return fmaxf(a, b); }
__device__ inline double max_r(double a, double b) { return fmax(a, b); }
__device__ inline float abs_r(float x) { return fabsf(x); }
__device__ inline double abs_r(double x) { return fabs(x); }
__device__ inline float pow_r(float x, float y) { return powf(x, y); }
// x^y for x <= 1 where only a smooth WEIGHT depends on it (never a direction or a discrete decision):
// v_exp_f32(y * v_log_f32(x)), ~1e-6 relative for y <= 100
// (x <= 0 happens when the half vector dips below the surface: keep libm's answer there, e.g. (-x)^30 > 0)
__device__ inline float pow_weight_r(float x, float y)
{
    return x > 0.f ? __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)) : powf(x, y);
}
__device__ inline double pow_weight_r(double x, double y) { return pow(x, y); }
__device__ inline double pow_r(double x, double y) { return pow(x, y); }
__device__ inline void sincospi_r(float x, float* s, float* c) { sincospif(x, s, c); }
__device__ inline void sincospi_r(double x, double* s, double* c) { sincospi(x, s, c); }
// sin and cos of phi = 2 pi u for the 31-bit draw r, f32: drt_sincos.h (also compiled on the host by its known-answer test)
#ifndef DRT_F64_LIBM
// f64 (the verification mode's bound is 1e-9 of the reference, not its last bit): u = r / RAND_MAX as a product with the
// reciprocal (one ulp; the IEEE division is ten instructions, three draws per vertex), and sin / cos of 2 pi u without the
// math library's sincospi (~70 instructions of general range reduction and special cases): the argument in QUARTER TURNS,
// x = 4 r / RAND_MAX in [0, 4], its nearest integer is the quadrant, the remainder times pi / 2 lies in [-pi/4, pi/4] -- exact up
// to the product's rounding, 1e-15 rad -- and Cephes' double-precision minimax polynomials take it from there (abs error
// <= 2e-15 against libm over all 2^31 draws' range, swept on the host: tests/cpp/sincos_kat.cpp).  Round 6: 467 -> ~380 vector
// instructions per wave-bounce of the f64 k_path.
__device__ inline double u01_f64(uint32_t r) { return (double)r * (1.0 / DRT_RAND_MAX_D); }   // random.hpp:9
__device__ inline void sincos_2pi_u31(uint32_t r, double* s, double* c) { drt_sincos_2pi_u31_f64(r, s, c); }
#else
__device__ inline double u01_f64(uint32_t r) { return (double)r / DRT_RAND_MAX_D; }   // random.hpp:9
__device__ inline void sincos_2pi_u31(uint32_t r, double* s, double* c) { sincospi(2.0 * u01_f64(r), s, c); }
#endif
template <typename R> __device__ inline V3<R> normalize(V3<R> a) { return a * rsqrt_r(dot(a, a)); }
// vector.hpp:602-606
template <typename R> __device__ inline V3<R> reflect(V3<R> v, V3<R> n) { return n * (R(2) * dot(n, v)) - v; }


// ---- shapes: shape.hpp:49-59 (Plane), 78-106 (Sphere) ----------------------------------------
// Same predicates as the reference (t > 0, NaN never hits); the caller keeps the first shape on
// ties (pathtracer.hpp:80).
template <typename R>
__device__ inline bool shape_intersect(const DevShape<R>& s, V3<R> o, V3<R> d, R& t)
{
    if (s.type == DRT_SHAPE_PLANE) {
        V3<R> n = mk<R>(s.p[0], s.p[1], s.p[2]);
        R h = dot(o, n) - s.p[3];
        t = div_r(h, -dot(d, n));
        return t > R(0);
    }
    // Sphere, shape.hpp:78-103 without divergent branches: with sqrt(disc) >= 0, t1 <= t2, so the
    // reference's cascade (both > 0: min; else t1 > 0: t1; else t2 > 0: t2) is "t1 if t1 > 0
    // else t2", accepted when disc >= 0 and that t > 0 (NaN fails every test, as it does there).
    V3<R> oc = o - mk<R>(s.p[0], s.p[1], s.p[2]);
    R b = R(2) * dot(oc, d);
    R c = dot(oc, oc) - s.p[3] * s.p[3];
    R disc = b * b - R(4) * c;
    R sq = sqrt_r(disc > R(0) ? disc : R(0));
    R t1 = (-b - sq) * R(0.5);
    R t2 = (-b + sq) * R(0.5);
    t = t1 > R(0) ? t1 : t2;
    return disc >= R(0) && t > R(0);
}

template <typename R>
__device__ inline V3<R> shape_normal(const DevShape<R>& s, V3<R> p)
{
    if (s.type == DRT_SHAPE_PLANE)
        return mk<R>(s.p[0], s.p[1], s.p[2]);           // as stored, not normalised
    return normalize(p - mk<R>(s.p[0], s.p[1], s.p[2]));
}

// bxdf.hpp:29-41 make_frame: Gram-Schmidt against e1 or e2, frame[2] = normal AS GIVEN
template <typename R>
__device__ inline void make_frame(V3<R> n, V3<R>& t, V3<R>& b)
{
    // e - (e . n) n for e = e1 or e2, whichever is less aligned with n: both candidates are formed from ONE
    // selected component s = n.x or n.y (a select, not a divergent branch): t = e - s n
    const bool use_x = abs_r(n.x) < abs_r(n.y);
    const R s = use_x ? n.x : n.y;
    t = normalize(mk<R>((use_x ? R(1) : R(0)) - n.x * s, (use_x ? R(0) : R(1)) - n.y * s, -n.z * s));
    b = normalize(cross(n, t));
}

// ---- triangle: two-sided Moller-Trumbore, hit iff t > 0 (oracle/ref_harness.cpp `Triangle`) ---
template <typename R>
__device__ inline bool tri_intersect(V3<R> v0, V3<R> e1, V3<R> e2, V3<R> o, V3<R> d, R& t)
{
    const V3<R> pvec = cross(d, e2);
    const R det = dot(e1, pvec);
    if (det == R(0))
        return false;
    const R inv = div_r(R(1), det);           // (f32: v_rcp)
    const V3<R> tvec = o - v0;
    const R u = dot(tvec, pvec) * inv;
    if (u < R(0) || u > R(1))
        return false;
    const V3<R> qvec = cross(tvec, e1);
    const R v = dot(d, qvec) * inv;
    if (v < R(0) || u + v > R(1))
        return false;
    t = dot(e2, qvec) * inv;
    return t > R(0);
}

// conservative slab test against a (padded) box, limited to (0, tmax]; tn = entry distance
template <typename R>
__device__ inline bool box_hit(V3<R> lo, V3<R> hi, V3<R> o, V3<R> inv_d, R tmax, R& tn)
{
    const R ax = (lo.x - o.x) * inv_d.x, bx = (hi.x - o.x) * inv_d.x;
    const R ay = (lo.y - o.y) * inv_d.y, by = (hi.y - o.y) * inv_d.y;
    const R az = (lo.z - o.z) * inv_d.z, bz = (hi.z - o.z) * inv_d.z;
    tn = max_r(max_r(min_r(ax, bx), min_r(ay, by)), max_r(min_r(az, bz), R(0)));
    const R tf = min_r(min_r(max_r(ax, bx), max_r(ay, by)), min_r(max_r(az, bz), tmax));
    return tn <= tf;
}
