// drt_chain.h -- the reference's UNBIASED integration operator (integrate.hpp:11-24, 39-52) on the queue wavefront: the
// adjoint rounds' per-path chain kernels (the one-launch form for analytic scenes is k_path_unbiased, drt_path.h).
#pragma once

#include "drt_kernels.h"
#include "drt_backward.h"

// ---- unbiased backward (integrate.hpp:11-24, 39-52): adjoint rounds --------------------------------
// The reference's IntegrateBackward, at the vertex where a gradient arrives, draws a FRESH direction,
// evaluates forward(sample) -- a whole new suffix path -- back-propagates grad / pdf through
// brdf * radiance * cos, and the recursion continues down the NEW path.  As a wavefront: every path
// keeps a CHAIN VERTEX (the incoming ray and its hit, depth r) and the gradient g arriving there.
// Round r:  k_adj_vertex  (E-gradient bookkeeping, fresh theta/phi, suffix ray queued at depth r+1; scenes with a mesh:
//                          also its analytic hit and the BVH walk's candidate lists, <TAIL>)
//           K2/K3 over depths r+1 .. D-1   (the ordinary bounce loop writes the suffix's tape)
//           the suffix's first ray + FINAL hit = the next chain vertex: saved by the shade launch of depth r+1, which
//           holds both (path-indexed)
//           k_adj_accumulate                (L' of the suffix from its tape, gradients of round r, g and chain vertex
//                                            of round r+1)
template <typename R>
struct ChainState {
    typename Q4<R>::T* cv_a;      // (o.xyz, d.x) of the ray that reached the chain vertex
    typename Q2<R>::T* cv_b;      // (d.y, d.z)
    HitRec<R>* cv_hit;            // its hit; prim = -2: chain finished
    typename Q4<R>::T* nx_a;      // the suffix's first ray / hit (saved after K2 at depth r+1)
    typename Q2<R>::T* nx_b;
    HitRec<R>* nx_hit;
    typename Q4<R>::T* g;         // (g.rgb, RNG path key bits)
    typename Q4<R>::T* w;         // (g3.rgb, bs) of the current round
    uint32_t* ids;                // colour | emission << 16 of the chain vertex
    uint32_t* ndraw;              // next unused draw of the path's stream
    uint32_t* dbase;              // draw base of the current suffix (index of theta at depth r+1)
};

// after the forward pass: seed, draw position and liveness of every path's chain
template <typename R>
__global__ void __launch_bounds__(DRT_BLOCK)
k_adj_init(BatchArgs a, const TapeRec<R>* __restrict__ tape, const uint32_t* __restrict__ nv,
           const float* __restrict__ adjoint, ChainState<R> cs)
{
    typedef typename Q4<R>::T R4;
    const size_t N = a.n_paths;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_paths; i += stride) {
        const int K = (int)nv[i];
        const uint32_t sl = i / a.Pb, pl = i - sl * a.Pb;
        const uint64_t path = (uint64_t)global_pixel(a, a.p0 + pl) * (uint64_t)a.spp + (uint64_t)(a.s0 + sl);
        const V3<R> g = path_seed<R>(a, adjoint, i);
        R4 o;
        o.x = g.x; o.y = g.y; o.z = g.z; o.w = pid_pack(R(0), (uint32_t)path);
        cs.g[i] = o;
        if (K <= 0) {
            HitRec<R> h;
            h.t = R(0);
            h.prim = -2;
            cs.cv_hit[i] = h;
            continue;
        }
        // draws the forward pass consumed: 2 camera, 2 per vertex with a BxDF, one roulette draw at
        // every depth in [min_bounces, K] the walk reached (absorbed, missed, or -- zero-length rays
        // never hit -- the continuation after a BxDF-less vertex), none at the depth cap
        const bool last_null = (tape[(size_t)(K - 1) * N + i].ids & 0xFFFFu) == DRT_ID_NONE;
        const int top = (K < a.depth_cap || a.cap_draws) ? K : a.depth_cap - 1;
        const int rr = top - a.min_bounces + 1;
        cs.ndraw[i] = 2u + 2u * (uint32_t)(K - (last_null ? 1 : 0)) + (uint32_t)(rr > 0 ? rr : 0);
    }
}

// round r, step 1: one wave per queue region (like K1) over the PATHS of the region
// TAIL (scenes with a mesh): like k_shade<TAIL>, the kernel intersects the ray it PRODUCES with the analytic shapes, writes the
// hit lane of depth r + 1 and appends the ray to its region's candidate list for the BVH walk -- no k_intersect pass over
// the suffix's first rays.
template <typename R, bool SPEC, bool TAIL = false>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 && SPEC) ? 4 : 1)
k_adj_vertex(BatchArgs a, int r, const DevScene<R>* __restrict__ sc, const R* __restrict__ params,
             ChainState<R> cs, const typename Q4<R>::T* __restrict__ tri_shade,
             typename Q4<R>::T* __restrict__ ray_a, typename Q2<R>::T* __restrict__ ray_b,
             uint2* __restrict__ ray_id, uint32_t* __restrict__ nv, uint32_t* __restrict__ counts_s,
             DevBvh<R> bvh_t, HitRec<R>* __restrict__ hit_next, uint32_t* __restrict__ cand,
             typename Q4<R>::T* __restrict__ cand_a, typename Q4<R>::T* __restrict__ cand_b, uint32_t* __restrict__ cand_count)
{
    typedef typename Q4<R>::T R4;
    __shared__ SceneLds<R> lds;
    __shared__ ProgLds s_prog;                  // TAIL, f32: the kind-sorted intersection program of the analytic shapes
    ProgRecs<0> recs;
    recs.lds = &s_prog;
    if (TAIL)
        stage_tail_program(s_prog, sc);
    stage_scene(lds, sc, params);
    if (TAIL && blockIdx.x == 0 && threadIdx.x < DRT_PULL_COUNTERS)     // the walk's list counters (it runs after this kernel)
        pull_counters(cand_count, a.n_regions)[threadIdx.x * DRT_PULL_STRIDE] = 0;
    const uint32_t w = grid_wave();
    if (w >= a.n_regions)
        return;
    uint32_t cand_running = 0;
    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t begin = w << a.region_shift;
    const uint32_t end = min(begin + a.region_size, a.n_paths);
    const int s = r + 1;                                        // depth of the suffix's first ray
    const R inv_pr = r >= a.min_bounces ? (R)(1.0 / (1.0 - a.absorb)) : R(1);
    uint32_t running = 0;
    for (uint32_t off = begin; off < end; off += DRT_WAVE) {
        const uint32_t i = off + lane;
        bool emit = false;
        R4 na;
        typename Q2<R>::T nb;
        uint2 nid;
        if (i < end) {
            const HitRec<R> h = cs.cv_hit[i];
            if (h.prim >= 0) {
                const R4 ra = cs.cv_a[i];
                const typename Q2<R>::T rb = cs.cv_b[i];
                const R4 gk = cs.g[i];
                const uint32_t key = pid_unpack(gk.w);
                const V3<R> o = mk<R>(ra.x, ra.y, ra.z), d = mk<R>(ra.w, rb.x, rb.y);
                const V3<R> P = o + d * h.t;
                V3<R> nrm;
                int material, emitter;
                uint32_t cparam;
                resolve_hit(lds, tri_shade, h.prim, P, nrm, material, emitter, cparam);
                const uint32_t eid = emitter >= 0 ? (uint32_t)lds.sc.emitter_param[emitter] : DRT_ID_NONE;
                R4 wrec;
                wrec.x = wrec.y = wrec.z = wrec.w = R(0);
                uint32_t cid = DRT_ID_NONE;
                nv[i] = (uint32_t)s;                            // no suffix vertices unless K3 says so
                if (material >= 0) {
                    const DevMaterial<R>& m = lds.sc.materials[material];
                    cid = cparam;
                    const uint32_t n = cs.ndraw[i];
                    V3<R> wo;
                    R q, bs;
                    sample_bxdf<R, SPEC>(m, nrm, d, rng_draw(a.rng_stream, key, n), rng_draw(a.rng_stream, key, n + 1), wo, q, bs);
                    const R c = dot(nrm, wo);
                    // seed of forward(sample).backward: (g / p) / pdf, then * cos (integrate.hpp:17,
                    // vector.hpp:457)
                    const R scale = inv_pr / q * c;
                    wrec.x = gk.x * scale; wrec.y = gk.y * scale; wrec.z = gk.z * scale; wrec.w = bs;
                    // trace() of the suffix at depth s: cap, then roulette (pathtracer.hpp:128)
                    uint32_t used = 2;
                    emit = s < a.depth_cap;
                    if (emit && s >= a.min_bounces) {
                        emit = !(rng_draw(a.rng_stream, key, n + 2) < a.rr_threshold);
                        used = 3;
                    }
                    cs.ndraw[i] = n + used;
                    cs.dbase[i] = n + used;                     // theta of depth s
                    const V3<R> no = P + wo * R(1e-3);          // pathtracer.hpp:99
                    na.x = no.x; na.y = no.y; na.z = no.z; na.w = wo.x;
                    nb.x = wo.y; nb.y = wo.z;
                    nid.x = i; nid.y = key;
                }
                cs.w[i] = wrec;
                cs.ids[i] = cid | (eid << 16);
            }
        }
        uint32_t n_emit;
        const uint32_t slot = begin + running + wave_rank(emit, n_emit);
        if (emit) {
            ray_a[slot] = na;
            ray_b[slot] = nb;
            ray_id[slot] = nid;
        }
        if (TAIL)
            tail_emit<R>(a, sc, recs, bvh_t, emit, slot, na, nb, w, cand_running, hit_next, cand, cand_a, cand_b);
        running += n_emit;
    }
    if (lane == 0) {
        counts_s[w] = running;
        if (TAIL)
            cand_count[w] = cand_running;
    }
}

// round r, last step: the gradients of the round, then the chain moves to the suffix's first vertex
template <typename R, int NP>
__global__ void __launch_bounds__(DRT_BLOCK)
k_adj_accumulate(BatchArgs a, int r, const DevScene<R>* __restrict__ sc, const R* __restrict__ params,
                 const TapeRec<R>* __restrict__ tape, const uint32_t* __restrict__ nv, ChainState<R> cs,
                 double* __restrict__ gpart, double* __restrict__ grad, int n_rows, int row_stride)
{
    typedef typename Q4<R>::T R4;
    constexpr bool SMALL = NP > 0;
    __shared__ SceneLds<R> lds;
    __shared__ double acc_d[NP > 0 ? 1 : DRT_LDS_PARAMS * 3];
    R (*acc)[DRT_BLOCK] = reinterpret_cast<R(*)[DRT_BLOCK]>(acc_d);
    __shared__ double red[DRT_BLOCK / DRT_WAVE][DRT_FAST_PARAMS * 3];
    GradAcc<R, NP> ga;
    ga.init(acc);
    stage_scene(lds, sc, params);

    const size_t N = a.n_paths;
    const int s = r + 1;
    const R inv_pr = r >= a.min_bounces ? (R)(1.0 / (1.0 - a.absorb)) : R(1);
    const R inv_p_rr = (R)(1.0 / (1.0 - a.absorb));
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_paths; i += stride) {
        HitRec<R> h = cs.cv_hit[i];
        if (h.prim < 0) {
            cs.nx_hit[i] = h;                                   // (cv and nx change places after this launch: stays finished)
            continue;
        }
        const uint32_t ids = cs.ids[i];
        const uint32_t cid = ids & 0xFFFFu, eid = ids >> 16;
        const R4 gk = cs.g[i];
        if (eid != DRT_ID_NONE)                                 // AddBackward: emission first
            ga.add(acc, grad, eid, mk<R>(gk.x, gk.y, gk.z) * inv_pr);
        bool go_on = false;
        if (cid != DRT_ID_NONE) {
            const R4 wr = cs.w[i];
            const int K = (int)nv[i];
            // L' of the round's suffix: its tape records of depths s .. K - 1, deepest first (read once, by the thread that
            // needs their sum -- round 2 had a pass of its own for it)
            V3<R> Lsuf = mk<R>(R(0), R(0), R(0));
            for (int k = K - 1; k >= s; --k) {
                const TapeRec<R> tr = tape[(size_t)k * N + i];
                const uint32_t tc = tr.ids & 0xFFFFu, te = tr.ids >> 16;
                const R inv_pk = k >= a.min_bounces ? inv_p_rr : R(1);
                V3<R> Lk = mk<R>(R(0), R(0), R(0));
                if (te != DRT_ID_NONE)
                    Lk = load_param<R, SMALL>(lds, params, (int)te) * inv_pk;
                if (tc != DRT_ID_NONE)
                    Lk = Lk + load_param<R, SMALL>(lds, params, (int)tc) * (Lsuf * tr.m);
                Lsuf = Lk;
            }
            const V3<R> g3 = mk<R>(wr.x, wr.y, wr.z);
            ga.add(acc, grad, cid, Lsuf * g3 * wr.w);    // MulBackward, brdf side
            if (K > s) {                                        // the suffix has a first vertex
                const V3<R> gn = load_param<R, SMALL>(lds, params, (int)cid) * wr.w * g3;   // radiance side
                R4 o = gk;
                o.x = gn.x; o.y = gn.y; o.z = gn.z;
                cs.g[i] = o;
                // (the chain's next vertex -- the suffix's first ray and its hit -- is in nx already, saved by the shade launch
                //  of depth s; the launcher lets cv and nx change places instead of 32 bytes per path being copied here)
                // draws the suffix consumed after its base (see k_adj_init)
                const bool last_null = (tape[(size_t)(K - 1) * N + i].ids & 0xFFFFu) == DRT_ID_NONE;
                const int top = (K < a.depth_cap || a.cap_draws) ? K : a.depth_cap - 1;
                const int first_rr = a.min_bounces > s + 1 ? a.min_bounces : s + 1;
                const int rr = top - first_rr + 1;
                cs.ndraw[i] = cs.dbase[i] + 2u * (uint32_t)(K - s - (last_null ? 1 : 0)) + (uint32_t)(rr > 0 ? rr : 0);
                go_on = true;
            }
        }
        if (!go_on) {
            h.prim = -2;
            cs.nx_hit[i] = h;
        }
    }

    flush_grad_block<R, NP>(ga, acc, red, gpart, n_rows, row_stride);
}
