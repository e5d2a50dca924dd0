// drt_path_mesh.h -- k_path_mesh: k_path for scenes WITH a triangle mesh (gfx950 / wave64).
//
// The queue wavefront takes a mesh scene through 8 x (k_shade<TAIL>: 100 B per segment through HBM, only to hand rays from
// one launch to the next) + 8 x (k_intersect_mesh: the BVH walk, a chain of dependent fetches that leaves the vector pipe
// idle half of the time).  Here a path never leaves its lane: ONE launch, a lane = a pixel that walks its samples one after
// the other like the regenerating k_path (drt_path.h), and every lane is a small state machine --
//     NEW   no path: the camera code starts the pixel's next sample                       Camera::sample, camera.hpp:51-60
//     HIT   a ray whose closest hit is final: vertex, BxDF sample, next ray               Pathtracer::scatter, pathtracer.hpp:91-115
//     WALK  a ray with its analytic hit that reaches the bounds of the mesh: BVH walk     Pathtracer::raycast, pathtracer.hpp:72-89
// A wave runs the step most of its lanes wait for: the vertex step when `shade_min` lanes hold a final hit (or no path),
// else the interior-node loop / the leaf step of the walk ("while-while", as in k_intersect_mesh) -- so the walk's stalls
// hide behind the other waves' vertex arithmetic instead of behind a second set of launches, and no ray, hit record, tape
// record or candidate list exists in memory: per path the launch moves 0 bytes, like k_path.
// Gradients: k_path's bounce counters (drt_path.h, Tangents) evaluated where a path meets a light -- the sums of the
// reference's backward functors (vector.hpp:418-484) and `m_grad += grad` (vector.hpp:185-188) in closed form.
// The analytic shapes are tested by the kind-sorted program in LDS (drt_prog.h; the mesh record is left out), the BVH is
// read as the walk kernel reads it (quantised 64-byte nodes, drt_kernels.h).  The traversal stack: DRT_MESH_LDS_STACK entries per lane in LDS, the (rare) deeper
// ones in a global overflow area -- 99.6 % of the rays never hold more than 8, and 30 KB of stack per block would cost a
// block per CU.
#pragma once

#include "drt_path.h"
#include "drt_walk.h"

#ifndef DRT_MESH_LDS_STACK
#define DRT_MESH_LDS_STACK 16
#endif
#ifndef DRT_MESH_MIN_BLOCKS
#define DRT_MESH_MIN_BLOCKS 4
#endif

enum { DRT_MS_DONE = 0, DRT_MS_NEW = 1, DRT_MS_HIT = 2, DRT_MS_WALK = 3 };

template <typename R, bool SPEC, int NP, int NC>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 && NP <= 4) ? (NP == DRT_NP_ANY ? DRT_MESH_MIN_BLOCKS - 1 : DRT_MESH_MIN_BLOCKS) : 1)   // (the general form's tables: 41 KB of LDS, three blocks per CU)
k_path_mesh(PathArgs a, const DevScene<R>* __restrict__ sc, const R* __restrict__ params, const float* __restrict__ adjoint,
            DevBvh<R> bvh, uint32_t* __restrict__ ovf, uint32_t ovf_stride,
            double* __restrict__ gpart, double* __restrict__ fpart, uint32_t* __restrict__ counts,
            unsigned long long* __restrict__ total, double* __restrict__ gimg_part)
{
    if (total && blockIdx.x == 0 && threadIdx.x < 8)
        total[threadIdx.x] = 0;                           // (the finishing kernel behind this launch adds into them)
    typedef typename Q4<R>::T R4;
    typedef typename Q2<R>::T R2;
    __shared__ PathSceneLds<R> lds;
    __shared__ double s_red[DRT_BLOCK / DRT_WAVE][DRT_FAST_PARAMS * 3];
    __shared__ TangentLds<R> s_tl;
    __shared__ ProgLds s_prog;                            // f32: the kind-sorted program of the analytic shapes
    __shared__ uint32_t s_stack[DRT_MESH_LDS_STACK][DRT_BLOCK];
    __shared__ R s_acc[NP > 0 ? NP * 3 : 1][DRT_BLOCK];
    __shared__ uint32_t s_ih[DRT_DRAW_TABLE];             // h(n) of every draw index (lanes stand at their own depths: drt_path.h)
    constexpr bool GEN = NP == DRT_NP_ANY;                // any number of parameters: vertex history + per-wave tables (drt_path.h)
    __shared__ typename PickT<GEN, GenBlock<R>, NoLds>::T s_gen;
    extern __shared__ uint32_t s_hist[];                  // GEN: [a.hist_lds][DRT_BLOCK] history words
    for (uint32_t n = threadIdx.x; n < DRT_DRAW_TABLE; n += DRT_BLOCK)
        s_ih[n] = drt_rng_index_hash(a.rng_stream, n);
    if constexpr (GEN)
        gen_zero(s_gen);
    stage_tail_program(s_prog, sc);
    stage_path_scene(lds, sc, params);                    // (ends with a barrier)
    const TangentLds<R>& tl = s_tl;
    if (NC > 0 && !GEN)
        stage_tangents(s_tl, lds);
    ProgRecs<0> recs;
    recs.lds = &s_prog;

    const uint32_t tid = threadIdx.x, gtid = blockIdx.x * DRT_BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t w = grid_wave();                       // wave of the grid = group + n_groups * range
    const uint32_t range = w / a.n_groups, group = w - range * a.n_groups;
    const uint32_t lp = group * DRT_WAVE + lane;          // batch-local pixel of this lane
    const bool have = range < a.n_ranges && lp < a.Pb;
    const uint32_t s_begin = range * a.spr;
    const uint32_t s_end = s_begin + a.spr < a.Sb ? s_begin + a.spr : a.Sb;

    Tangents<R, NP, NC> tg;
    tg.acc = &s_acc[0][threadIdx.x];
    if constexpr (GEN)
        gen_begin(s_gen, lds, sc, a, s_hist, reinterpret_cast<uint32_t*>(a.hist_ovf), tg);   // (behind the traversal stacks' global part)
    else {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            tg.acc_set(p, mk<R>(R(0), R(0), R(0)));
    }
    double fx = 0, fy = 0, fz = 0;                        // radiance sum of this lane's pixel over the range
    uint32_t n_seg = 0, n_capped = 0, n_walked = 0;       // wave-uniform counters

    uint32_t gpix = 0, px = 0, py = 0;
    V3<R> g = mk<R>(R(1), R(1), R(1));                    // render.cpp:80: radiance.backward(Vec3(1))
    if (have) {
        gpix = path_global_pixel(a, a.p0 + lp);
        py = gpix / (uint32_t)a.W;
        px = gpix - py * (uint32_t)a.W;
        if (NP != 0 && adjoint)
            g = mk<R>((R)adjoint[(size_t)gpix * 3], (R)adjoint[(size_t)gpix * 3 + 1], (R)adjoint[(size_t)gpix * 3 + 2]);
    }
    CameraLane<R> cl;                                     // (see k_path)
    cl.cs0 = (R)((2. * (double)px * a.inv_W - 1.) * a.aspect * a.tan_half);
    cl.ct0 = (R)((2. * (double)py * a.inv_H - 1.) * a.tan_half);
    const R pk_rr = (R)a.p_rr, inv_p_rr = (R)a.inv_p_rr;
    const int n_shapes = lds.sc.n_shapes;
    const int first_rr = a.min_bounces > 1 ? a.min_bounces : 1;
    const V3<R> blo = mk<R>(bvh.lo[0], bvh.lo[1], bvh.lo[2]), bhi = mk<R>(bvh.hi[0], bvh.hi[1], bvh.hi[2]);

    // ---- the lane's path
    uint32_t st = (have && s_begin < s_end) ? DRT_MS_NEW : DRT_MS_DONE;
    uint32_t sl = s_begin, key = 0;                       // the lane's next sample; RNG key of its current path
    int kk = 0;                                           // depth of the vertex the current ray leads to
    V3<R> o = mk<R>(R(0), R(0), R(0)), d = o, inv_d = o;
    V3<R> T = mk<R>(R(1), R(1), R(1));
    R tmin = (R)INFINITY;
    int prim = -1, sp = 0;
    uint32_t best_flat = 0xFFFFFFFFu, cur = DRT_BVH_NONE;
    if (NC > 0 || GEN)
        tg.new_path();

#define DRT_MESH_PUSH(v)                                                                      \
    do {                                                                                      \
        if (sp < DRT_MESH_LDS_STACK) s_stack[sp][tid] = (v);                                  \
        else ovf[(size_t)(sp - DRT_MESH_LDS_STACK) * ovf_stride + gtid] = (v);                \
        ++sp;                                                                                 \
    } while (0)
#define DRT_MESH_POP(dst)                                                                     \
    do {                                                                                      \
        if (sp > 0) {                                                                         \
            --sp;                                                                             \
            if (sp < DRT_MESH_LDS_STACK) dst = s_stack[sp][tid];                              \
            else dst = ovf[(size_t)(sp - DRT_MESH_LDS_STACK) * ovf_stride + gtid];            \
        } else {                                                                              \
            dst = DRT_BVH_NONE;                                                               \
            st = DRT_MS_HIT;                              /* the walk is over: (tmin, prim) is the ray's closest hit */ \
        }                                                                                     \
    } while (0)

    if (range < a.n_ranges)
    for (;;) {
        const uint64_t m_ready = wave_ballot(st == DRT_MS_HIT || st == DRT_MS_NEW);
        const uint64_t m_walk = wave_ballot(st == DRT_MS_WALK);
        if ((m_ready | m_walk) == 0)
            break;                                        // every lane is through its samples
        if ((uint32_t)__popcll(m_ready) >= a.shade_min || m_walk == 0) {
            bool fresh = false;                           // the lane has a new ray to intersect
            bool capped = false;                          // a user max_depth (not the roulette) cut the lane's path short
            // ---- vertices: lanes whose ray has its final hit (pathtracer.hpp:91-115, 128-133)
            if (st == DRT_MS_HIT) {
                const bool hit = prim >= 0;
                const V3<R> P = o + d * tmin;             // pathtracer.hpp:83
                V3<R> nrm;
                uint32_t ids;
                int material;
                if (prim >= n_shapes) {                   // a triangle: its record comes from L2
                    const R4 ts = bvh.tri_shade[prim - n_shapes];
                    const uint32_t wd = pid_unpack(ts.w);
                    const uint32_t mat = (wd >> 16) & 0xFFu, em = wd >> 24;       // (colour parameter | material << 16 | emitter << 24)
                    nrm = mk<R>(ts.x, ts.y, ts.z);
                    material = mat == 0xFFu ? 0 : (int)mat;
                    const uint32_t e_id = em == 0xFFu ? DRT_ID_NONE : (uint32_t)lds.sc.emitter_param[em];
                    ids = (wd & 0xFFFFu) | (e_id << 16);
                } else {
                    const DevShape<R>& sh = lds.sc.shapes[hit ? prim : 0];   // (a miss reads record 0, uses nothing of it)
                    ids = (uint32_t)sh.pad;
                    material = sh.material;
                    const V3<R> ctr = mk<R>(sh.p[0], sh.p[1], sh.p[2]);
                    const V3<R> nsph = normalize(P - ctr);                  // shape.hpp:105-106
                    const bool is_plane = sh.type == DRT_SHAPE_PLANE;       // shape.hpp:58-59: the normal as stored
                    nrm = mk<R>(is_plane ? ctr.x : nsph.x, is_plane ? ctr.y : nsph.y, is_plane ? ctr.z : nsph.z);
                }
                const uint32_t cid = ids & 0xFFFFu, eid = ids >> 16;
                const bool has_bxdf = cid != DRT_ID_NONE, emits = hit && eid != DRT_ID_NONE;
                const bool rr_here = kk >= a.min_bounces;
                const R pk = rr_here ? pk_rr : R(1);                        // pathtracer.hpp:130
                const R inv_pk = rr_here ? inv_p_rr : R(1);
                // emission (pathtracer.hpp:113-114), with or without a BxDF: radiance to the pixel, gradients to the lane's sums
                if (wave_any(emits)) {
                    if (emits) {
                        V3<R> Lc = mk<R>(R(0), R(0), R(0));
                        add_emission<R, NP, NC>(lds, tl, params, eid, inv_pk, T, g, Lc, tg);
                        fx += (double)Lc.x; fy += (double)Lc.y; fz += (double)Lc.z;
                    }
                }
                // the BxDF: sample, evaluate (pathtracer.hpp:91-111); draws of this depth
                const int rr_draws = kk - first_rr + 1;
                const uint32_t n_theta = 2u * (uint32_t)kk + (uint32_t)(rr_draws > 0 ? rr_draws : 0) + camera_draw_base(a.min_bounces);
                const DevMaterial<R>& m = lds.sc.materials[has_bxdf ? material : 0];
                V3<R> wo;
                R q, bs;
                sample_bxdf<R, SPEC>(m, nrm, d, drt_rng_combine(s_ih[n_theta], key), drt_rng_combine(s_ih[n_theta + 1], key), wo, q, bs);
                const R c = dot(nrm, wo);                                   // pathtracer.hpp:103
                const R mk_ = div_r(bs * c, q * pk);                        // T_{k+1} = T_k * colour * m_k
                const bool next_rr = (kk + 1) >= a.min_bounces, next_cap = (kk + 1) >= a.depth_cap;
                const bool rr_kills = next_rr && drt_rng_combine(s_ih[n_theta + 2], key) < a.rr_threshold;   // pathtracer.hpp:128
                const bool alive = hit && has_bxdf && !next_cap && !rr_kills;
                capped = hit && has_bxdf && next_cap && !rr_kills;
                const int cidx = has_bxdf ? (int)cid : 0;
                V3<R> col;
                if constexpr (GEN) {
                    const R* rec = tg.gl->colnz[cidx < DRT_PATH_LDS_PARAMS ? cidx : 0];
                    col = mk<R>(rec[0], rec[1], rec[2]);
                    tg.zc += pid_unpack(rec[3]);
                    tg.template push<false>(alive, cid);
                } else
                    col = NC > 0 ? mk<R>(tl.colnz[cidx][0], tl.colnz[cidx][1], tl.colnz[cidx][2]) : load_param<R, (NP > 0)>(lds, params, cidx);
                T = T * col * mk_;                                          // (only read again if the path goes on)
                if constexpr (NC > 0 && !GEN) {
                    tg.cnt[0] += tl.inc[cidx][0];
                    if (NC > 4)
                        tg.cnt[NC > 4 ? 1 : 0] += tl.inc[cidx][1];
                    tg.zc += tl.inc[cidx][2];
                }
                o = P + wo * R(1e-3);                                       // pathtracer.hpp:99
                d = wo;
                ++kk;
                fresh = alive;
                if (!alive)
                    st = sl < s_end ? DRT_MS_NEW : DRT_MS_DONE;
            }
            if (!a.cap_is_roulette)
                n_capped += (uint32_t)__popcll(wave_ballot(capped));
            // ---- lanes without a path start their next sample (camera.hpp:51-60) -- once enough of them wait, or as many as
            // still trace
            {
                const bool start = st == DRT_MS_NEW;
                const uint32_t n_start = (uint32_t)__popcll(wave_ballot(start));
                const uint32_t n_other = (uint32_t)__popcll(wave_ballot(fresh)) + (uint32_t)__popcll(m_walk);
                if (n_start >= a.regen_min || (n_start > 0 && n_start >= n_other)) {
                    if (start) {
                        R4 ra;
                        R2 rb;
                        key = path_camera<R, false>(a, cl, gpix, px, py, sl, ra, rb);   // (camera code inside the loop, scalar registers all taken: see path_camera)
                        ++sl;
                        kk = 0;
                        o = mk<R>(ra.x, ra.y, ra.z);
                        d = mk<R>(ra.w, rb.x, rb.y);
                        T = mk<R>(R(1), R(1), R(1));
                        if (NC > 0 || GEN)
                            tg.new_path();
                        // pathtracer.hpp:128 at depth 0
                        fresh = a.depth_cap > 0 && !(a.min_bounces <= 0 && rng_draw(a.rng_stream, key, 2) < a.rr_threshold);
                        if (!fresh)
                            st = sl < s_end ? DRT_MS_NEW : DRT_MS_DONE;
                    }
                }
            }
            // ---- the new rays: closest analytic shape; the ones that reach the bounds of the mesh before it go on to the walk
            n_seg += (uint32_t)__popcll(wave_ballot(fresh));
            bool reach = false;
            if (fresh) {
                R4 ra;
                R2 rb;
                ra.x = o.x; ra.y = o.y; ra.z = o.z; ra.w = d.x;
                rb.x = d.y; rb.y = d.z;
                const HitRec<R> hn = tail_closest_hit(sc, recs, ra, rb);
                tmin = hn.t;
                prim = hn.prim;
                inv_d = mk<R>(div_r(R(1), d.x), div_r(R(1), d.y), div_r(R(1), d.z));   // (f32: v_rcp; the boxes are padded)
                R tn;
                reach = box_hit(blo, bhi, o, inv_d, tmin, tn);
                st = reach ? DRT_MS_WALK : DRT_MS_HIT;
                if (reach) {
                    best_flat = prim >= 0 ? (uint32_t)lds.sc.flat[prim] : 0xFFFFFFFFu;
                    cur = 0;                                                // root
                    sp = 0;
                }
            }
            n_walked += (uint32_t)__popcll(wave_ballot(reach));
        }
        if (!wave_any(st == DRT_MS_WALK))
            continue;
        // ---- the walk.  Interior nodes: tight loop, leaves postponed; left as soon as too few lanes still descend and the
        // others have something to do (a leaf, or enough final hits for a vertex step)
        for (;;) {
            const bool descending = st == DRT_MS_WALK && !(cur & DRT_BVH_LEAF);
            const uint64_t dmask = wave_ballot(descending);
            if (dmask == 0)
                break;
            if ((uint32_t)__popcll(dmask) < a.descend_min) {
                // (few walkers in the wave: the loop goes on while at least half of THEM descend)
                const uint64_t lmask = wave_ballot(st == DRT_MS_WALK && (cur & DRT_BVH_LEAF));
                if (lmask != 0 && __popcll(dmask) < __popcll(lmask))
                    break;
                if ((uint32_t)__popcll(wave_ballot(st == DRT_MS_HIT || st == DRT_MS_NEW)) >= a.shade_min)
                    break;
            }
            if (!descending)
                continue;
            R tc[4];
            uint32_t lc[4];
            const uint4* np_ = bvh.node + (size_t)cur * 4;
            const uint4 w0 = np_[0], w1 = np_[1], w2 = np_[2], w3 = np_[3];
            quant_node_visit<R>(w0, w1, w2, w3, o, inv_d, tmin, tc, lc);
            sort4_by_t<R>(tc, lc);
            // farthest first onto the stack, nearest becomes current
            if (tc[3] < (R)INFINITY) DRT_MESH_PUSH(lc[3]);
            if (tc[2] < (R)INFINITY) DRT_MESH_PUSH(lc[2]);
            if (tc[1] < (R)INFINITY) DRT_MESH_PUSH(lc[1]);
            if (tc[0] < (R)INFINITY)
                cur = lc[0];
            else
                DRT_MESH_POP(cur);
        }
        // ---- leaves
        if (st == DRT_MS_WALK && (cur & DRT_BVH_LEAF)) {
            leaf_visit<R>(bvh.tri, cur, n_shapes, o, d, tmin, prim, best_flat);
            DRT_MESH_POP(cur);
        }
    }
#undef DRT_MESH_POP
#undef DRT_MESH_PUSH

    if (range < a.n_ranges) {
        if (fpart && have) {
            double* f = fpart + ((size_t)range * 3) * a.Pb + lp;       // [range][channel][pixel]: coalesced
            f[0] = fx; f[(size_t)a.Pb] = fy; f[(size_t)a.Pb * 2] = fz;
        }
        if constexpr (NP > 0) if (gimg_part && have) {                 // gradient image (README.md:142-145): see k_path
            const V3<R> v = tg.acc_get(a.gimg_param > 0 && a.gimg_param < NP ? a.gimg_param : 0);
            double* f = gimg_part + ((size_t)range * 3) * a.Pb + lp;
            f[0] = (double)v.x; f[(size_t)a.Pb] = (double)v.y; f[(size_t)a.Pb * 2] = (double)v.z;
        }
        if (lane == 0) {
            const size_t nw = (size_t)a.n_groups * a.n_ranges;
            counts[w] = n_seg;
            counts[nw + w] = n_capped;
            counts[2 * nw + w] = n_walked;
        }
    }
    if constexpr (GEN)
        gen_finish(s_gen, a, gpart);
    else
    if (NP > 0) {
        // block reduction in fp64: thread -> wave (shuffles) -> block (LDS), fixed order; the finishing launch adds the blocks
        const int wv = threadIdx.x / DRT_WAVE;
#pragma unroll
        for (int r = 0; r < NP * 3; ++r) {
            double v = (double)tg.acc[r * DRT_BLOCK];
#pragma unroll
            for (int o2 = DRT_WAVE / 2; o2 > 0; o2 >>= 1)
                v += __shfl_down(v, o2);
            if (lane == 0)
                s_red[wv][r] = v;
        }
        __syncthreads();
        if (threadIdx.x < DRT_FAST_PARAMS * 3) {
            double v = 0;
            if ((int)threadIdx.x < NP * 3)
                for (int ww = 0; ww < DRT_BLOCK / DRT_WAVE; ++ww)
                    v += s_red[ww][threadIdx.x];
            gpart[(size_t)blockIdx.x * (DRT_FAST_PARAMS * 3) + threadIdx.x] = v;
        }
    }
}
