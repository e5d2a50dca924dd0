// drt_prog.h -- the "intersection program" of the analytic shapes: one record per shape with a KIND (general plane,
// axis plane, sphere), tested either from scalar registers with the kinds compiled in (k_path on the reference's own
// scene) or from a kind-sorted copy in LDS (any other scene; also k_shade's tail in scenes with a mesh, where the
// program covers the analytic shapes and the mesh records are left out).  Built by drt_hip_upload_scene (fill_scene).
#pragma once

#include "drt_device.h"

// ---- the intersection program (f32) ----------------------------------------------------------------
// (two cheaper forms of this accept were measured on config 3 and not kept: HISTORY.md 3a)
// (f64, the verification mode: the same formulas with a full-precision reciprocal and square root; only the compiled-in
// form -- the reference's own scene -- exists in f64, every other scene keeps the literal loop there)
__device__ inline float prog_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline double prog_rcp(double x) { return div_r(1.0, x); }
__device__ inline float prog_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ inline double prog_sqrt(double x) { return sqrt_r(x); }

// ---- caller-defined kinds (DRT_SHAPE_USER, include/drt_hip.h: drt_shape_kind_desc) ------------------------------------------
// Their intersect / normal bodies exist only in a kernel hiprtc compiles for the scene: drt_jit.h hands the compiler a header
// "drt_user_shapes.h" made from the scene's sources --
//     template <typename R> __device__ inline bool drt_user_intersect_0(const R* p, V3<R> o, V3<R> d, R& t) { ... }
//     template <typename R> __device__ inline V3<R> drt_user_normal_0(const R* p, V3<R> P) { ... }       (and _1)
// -- and defines DRT_USER_SHAPES.  The library's own build knows no such kind: a scene that holds one renders through its own
// kernel or not at all (DRT_ERR_UNSUPPORTED).
#ifdef DRT_USER_SHAPES
#include "drt_user_shapes.h"
template <int K, typename R>
__device__ inline bool user_intersect(const R* p, V3<R> o, V3<R> d, R& t)
{
    if (K == 0) return drt_user_intersect_0<R>(p, o, d, t);
    return drt_user_intersect_1<R>(p, o, d, t);
}
template <typename R>
__device__ inline V3<R> user_normal(int kind, const R* p, V3<R> P)
{
    if (kind == 0) return drt_user_normal_0<R>(p, P);
    return drt_user_normal_1<R>(p, P);
}
// ... and caller-defined BxDF kinds (drt_bxdf_kind_desc): sample + evaluate in one body
template <typename R>
__device__ inline void user_bxdf(int kind, const R* p, V3<R> n, V3<R> d, R u1, R u2, V3<R>& wo, R& pdf, R& bs)
{
    if (kind == 0) drt_user_bxdf_0<R>(p, n, d, u1, u2, wo, pdf, bs);
    else drt_user_bxdf_1<R>(p, n, d, u1, u2, wo, pdf, bs);
}
#endif

template <typename R>
__device__ inline void prog_accept(R t, int s, R& tmin, int& prim)
{
    if (t > R(0) && !(t >= tmin)) {          // shape.hpp:55 / pathtracer.hpp:80: first shape wins ties
        tmin = t;
        prim = s;
    }
}

// t of one record, every product-sum written as explicit fma chains: the kind-sorted program evaluates a record a
// second time when it has seen an exact tie (below), and both evaluations must agree in all 32 bits whatever the
// compiler's contraction choices would have been.  Returns false where the reference's predicate cannot hold
// (sphere: negative discriminant).
template <int KIND, typename R>
__device__ inline bool prog_t(const typename Q4<R>::T r, V3<R> o, V3<R> d, V3<R> inv_d, R& t)
{
    if (KIND == DRT_PK_AX) {
        t = (r.x - o.x) * inv_d.x;
    } else if (KIND == DRT_PK_AY) {
        t = (r.x - o.y) * inv_d.y;
    } else if (KIND == DRT_PK_AZ) {
        t = (r.x - o.z) * inv_d.z;
    } else if (KIND == DRT_PK_PLANE) {
        const R h = fma_r(o.z, r.z, fma_r(o.y, r.y, o.x * r.x)) - r.w;
        const R den = fma_r(d.z, r.z, fma_r(d.y, r.y, d.x * r.x));
        t = h * prog_rcp(-den);
    } else {                                 // sphere: b' = oc.d, disc' = b'^2 - (oc.oc - r^2), t = -b' -+ sqrt(disc')
        const V3<R> oc = mk<R>(o.x - r.x, o.y - r.y, o.z - r.z);
        const R bh = fma_r(oc.z, d.z, fma_r(oc.y, d.y, oc.x * d.x));
        const R cc = fma_r(-r.w, r.w, fma_r(oc.z, oc.z, fma_r(oc.y, oc.y, oc.x * oc.x)));
        const R disc = fma_r(bh, bh, -cc);
        if (!(disc >= R(0)))                 // (NaN fails, as in the reference; the branch-free form -- sqrt of a negative
            return false;                    // = NaN, which fails `t > 0` -- was measured 4 % slower)
        const R sq = prog_sqrt(disc);
        const R t1 = -bh - sq, t2 = sq - bh;
        t = t1 > R(0) ? t1 : t2;
    }
    return true;
}

template <int KIND, typename R>
__device__ inline void prog_test(const typename Q4<R>::T r, int s, V3<R> o, V3<R> d, V3<R> inv_d, R& tmin, int& prim)
{
    R t;
    if (prog_t<KIND, R>(r, o, d, inv_d, t))
        prog_accept(t, s, tmin, prim);
}

// the kind-sorted program's version: also notes an exact tie with the closest hit so far (see closest_hit_prog)
template <int KIND>
__device__ inline void prog_test_tie(const float4 r, int s, V3<float> o, V3<float> d, V3<float> inv_d, float& tmin, int& prim, bool& tie)
{
    float t;
    if (prog_t<KIND, float>(r, o, d, inv_d, t)) {
        tie = tie || t == tmin;
        prog_accept(t, s, tmin, prim);
    }
}

// second pass after a tie: among the shapes whose t equals the closest hit bit for bit, the FIRST IN SCENE ORDER wins
template <int KIND>
__device__ inline void prog_resolve_tie(const float4 r, int s, V3<float> o, V3<float> d, V3<float> inv_d, float tmin, int& prim)
{
    float t;
    if (prog_t<KIND, float>(r, o, d, inv_d, t) && t == tmin && s < prim)
        prim = s;
}

// The kinds of a scene's shapes as a TYPE (3 bits per shape, 16 shapes per word, scene order): the template argument of
// the compiled-in program.  n == 0: no signature -- the kind-sorted program in LDS.  The reference's own scene is
// instantiated in the library (SigCornell); every other analytic scene gets its instantiation at run time (drt_jit.h:
// hiprtc compiles k_path<..., KindSig<the scene's words>, ...> when the scene has rendered enough to pay for it).
template <unsigned long long W0, unsigned long long W1, unsigned long long W2, unsigned long long W3, int N>
struct KindSig {
    static constexpr int n = N;
    static __host__ __device__ constexpr int kind(int s)
    {
        return (int)(((s < 16 ? W0 : s < 32 ? W1 : s < 48 ? W2 : W3) >> (3 * (s & 15))) & 7ull);
    }
    // where shape s stands in the scene's kind-sorted record array (DevScene::sorted: kinds ascending, scene order kept
    // inside a kind) -- a compile-time constant, so the compiled-in program and the kind-sorted one share ONE copy of the records
    static __host__ __device__ constexpr int pos(int s)
    {
        int p = 0;
        for (int i = 0; i < N; ++i)
            if (kind(i) < kind(s) || (kind(i) == kind(s) && i < s))
                ++p;
        return p;
    }
};
typedef KindSig<0ull, 0ull, 0ull, 0ull, 0> SigNone;

// The records of the program, as the bounce loop sees them.  SG::n > 0 (kinds fixed at compile time): the records are
// loaded ONCE per wave, before the sample loop, and stay in scalar registers -- the loop body then contains no scalar
// load, no wait and no branch for the scene at all.  SG::n == 0: the kind-sorted copy in LDS.
struct ProgLds {               // the kind-sorted program in LDS (scenes whose kinds are not compiled in)
    float4 rec[DRT_PROG_SORTED_MAX];
    int shape[DRT_PROG_SORTED_MAX];
    int kind_begin[8];
};

template <int NSIG, typename R = float>
struct ProgRecs {
    typename Q4<R>::T r[NSIG > 0 ? NSIG : 1];
#ifdef DRT_USER_SHAPES
    typename Q4<R>::T q[NSIG > 0 ? NSIG : 1];      // values 4..7 of a caller-defined shape's record (the others' entries are never read)
#endif
    const ProgLds* lds;
    template <typename SG>
    __device__ inline void load(const DevScene<R>* __restrict__ sc)
    {
#pragma unroll
        for (int s = 0; s < NSIG; ++s) {
            r[s] = *reinterpret_cast<const typename Q4<R>::T*>(sc->sorted[SG::pos(s)]);
#ifdef DRT_USER_SHAPES
            if (SG::kind(s) >= DRT_PK_USER0 && SG::kind(s) < DRT_PK_USER0 + DRT_MAX_USER_KINDS)
                q[s] = *reinterpret_cast<const typename Q4<R>::T*>(sc->user_q[s]);
#endif
        }
    }
};

// the compiled-in form, f32 or f64: SG::n records in scalar registers, kinds from SG
template <typename SG, typename R>
__device__ inline HitRec<R> closest_hit_sig(const ProgRecs<SG::n, R>& recs, V3<R> o, V3<R> d)
{
    const V3<R> inv_d = mk<R>(prog_rcp(d.x), prog_rcp(d.y), prog_rcp(d.z));
    R tmin = (R)INFINITY;
    int prim = -1;
#pragma unroll
    for (int s = 0; s < SG::n; ++s) {
        const int kind = SG::kind(s);
        const typename Q4<R>::T r = recs.r[s];
        if (kind == DRT_PK_AX) prog_test<DRT_PK_AX, R>(r, s, o, d, inv_d, tmin, prim);
        else if (kind == DRT_PK_AY) prog_test<DRT_PK_AY, R>(r, s, o, d, inv_d, tmin, prim);
        else if (kind == DRT_PK_AZ) prog_test<DRT_PK_AZ, R>(r, s, o, d, inv_d, tmin, prim);
        else if (kind == DRT_PK_PLANE) prog_test<DRT_PK_PLANE, R>(r, s, o, d, inv_d, tmin, prim);
#ifdef DRT_USER_SHAPES
        else if (kind >= DRT_PK_USER0 && kind < DRT_PK_USER0 + DRT_MAX_USER_KINDS) {
            const typename Q4<R>::T q = recs.q[s];
            const R p8[8] = {r.x, r.y, r.z, r.w, q.x, q.y, q.z, q.w};
            R t;
            if (kind == DRT_PK_USER0 ? user_intersect<0, R>(p8, o, d, t) : user_intersect<1, R>(p8, o, d, t))
                prog_accept(t, s, tmin, prim);
        }
#endif
        else prog_test<DRT_PK_SPHERE, R>(r, s, o, d, inv_d, tmin, prim);
    }
    HitRec<R> h;
    h.t = tmin;
    h.prim = prim;
    return h;
}

template <typename SG>
__device__ inline HitRec<float> closest_hit_prog(const DevScene<float>* __restrict__ sc, const ProgRecs<SG::n>& recs,
                                                 V3<float> o, V3<float> d)
{
    (void)sc;
    if (SG::n > 0)
        return closest_hit_sig<SG, float>(recs, o, d);
    const V3<float> inv_d = mk<float>(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    float tmin = INFINITY;
    int prim = -1;
    {
        // kinds not compiled in: the records sorted by kind (upload), one counted loop per kind -- no branch on the kind,
        // no scalar load: every lane reads the SAME record from LDS (a broadcast read), the next record is requested
        // while the current one is tested.  Inside a kind the scene order is kept, but an exact tie between shapes of
        // DIFFERENT kinds (two formulas agreeing in all 32 bits) would go to the kind tested first instead of the
        // earlier shape (pathtracer.hpp:80).  So every test also notes whether its t EQUALS the closest hit so far
        // (one compare): any shape that ties with the final winner is seen that way -- tested after the winner it meets
        // tmin == t, tested before it IS the winner and the later one meets it -- and a wave in which some lane saw a tie
        // (in practice never) walks the records once more, giving the hit to the smallest shape index among the equals.
        const ProgLds& pl = *recs.lds;
        bool tie = false;
#define DRT_KIND_LOOP(K)                                                              \
        for (int i = pl.kind_begin[K]; i < pl.kind_begin[K + 1]; ++i) {               \
            const float4 r = pl.rec[i];                                               \
            prog_test_tie<K>(r, pl.shape[i], o, d, inv_d, tmin, prim, tie);           \
        }
        DRT_KIND_LOOP(DRT_PK_AX)
        DRT_KIND_LOOP(DRT_PK_AY)
        DRT_KIND_LOOP(DRT_PK_AZ)
        DRT_KIND_LOOP(DRT_PK_PLANE)
        DRT_KIND_LOOP(DRT_PK_SPHERE)
#undef DRT_KIND_LOOP
        if (__any(tie && prim >= 0)) {
#define DRT_KIND_LOOP(K)                                                              \
            for (int i = pl.kind_begin[K]; i < pl.kind_begin[K + 1]; ++i) {           \
                const float4 r = pl.rec[i];                                           \
                prog_resolve_tie<K>(r, pl.shape[i], o, d, inv_d, tmin, prim);         \
            }
            DRT_KIND_LOOP(DRT_PK_AX)
            DRT_KIND_LOOP(DRT_PK_AY)
            DRT_KIND_LOOP(DRT_PK_AZ)
            DRT_KIND_LOOP(DRT_PK_PLANE)
            DRT_KIND_LOOP(DRT_PK_SPHERE)
#undef DRT_KIND_LOOP
        }
    }
    HitRec<float> h;
    h.t = tmin;
    h.prim = prim;
    return h;
}

