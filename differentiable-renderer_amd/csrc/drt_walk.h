// drt_walk.h -- K2 on triangles: the ordered walk of the 4-wide BVH (drt_bvh.h) that refines the analytic closest hit
// (Pathtracer::raycast, pathtracer.hpp:72-89, over a mesh's triangles), as a kernel of its own for the queue wavefront
// (k_intersect_mesh) and as the node / leaf steps k_path_mesh (drt_path_mesh.h) runs inside its one launch.
// What was measured on it and not kept is in HISTORY.md (sections 3c and "round 5").
#pragma once

#include "drt_kernels.h"

// ---- K2 (scenes with triangle meshes) -----------------------------------------------------------
// One closest-hit query = the analytic shapes (done by k_intersect at full lane efficiency just
// before; this kernel starts from that hit record), then an ORDERED walk of the BVH (near child
// first, far child on a 32-entry per-lane stack in LDS).  The top of the tree is staged in LDS once per
// block; deeper nodes and the triangles come from L2.  Incoherent rays make traversal lengths
// wildly different from lane to lane, so the kernel is organised around keeping lanes busy:
//   * every wave owns a private STREAM of rays (its chunks of the address-ordered sweep); when
//     DRT_BVH_REFILL lanes have finished, they are refilled from the stream with a wave ballot +
//     prefix rank -- the same atomic-free wave-local trick as the queue regions;
//   * interior nodes are walked in a tight inner loop and leaves are postponed until the lanes
//     meet again ("while-while"), so triangle tests run with many lanes active.
// Exact ties keep the primitive that comes first in the flattened scene, like the reference's
// linear scan (pathtracer.hpp:80): (t, flat index) is compared lexicographically.
#ifdef DRT_BVH_STATS
// debug build only (tools/): [0] rays, [1] node visits served from LDS, [2] from memory, [3] leaf visits, [4] triangle tests
__device__ unsigned long long g_bvh_stats[16];   // [8] node visits / [9] leaf visits whose entry distance lies beyond the hit found meanwhile; [10..15] rays by deepest stack (<=4, 8, 12, 16, 24, more)
// [0..7]: rays by their number of node visits (1, 2, 3-4, 5-8, 9-16, 17-32, 33-64, more); [8..15]: those of them that ended
// on a triangle; [16..23]: node visits summed per bin
__device__ unsigned long long g_bvh_hist[24];
#define DRT_STAT(i, n) atomicAdd(&g_bvh_stats[i], (unsigned long long)(n))
#else
#define DRT_STAT(i, n)
#endif
#ifdef DRT_WALK_TIMES
// debug build only (tools/walk_diag.py): per wave of the LAST walk launch, s_memrealtime (100 MHz) at its start, when its
// list counters ran dry, at its exit
__device__ unsigned long long g_walk_times[8192][3];
#endif


// ---- one visit of a QUANTISED node (64 B, drt_bvh.h: QNode; its four words are in w0..w3) ----------------------------------
// Four slab tests in the node's own grid: a bound plane at origin + q * 2^e is crossed at
//   t = ((origin - o) + q * 2^e) / d = A + q * B,   A = (origin - o) * inv_d,  B = 2^e * inv_d
// -- one conversion and one fma per plane instead of decoding the box first (fma, sub, mul) -- and the
// sign of d says which of a child's two planes per axis is the near one, so no min / max pairs either.
// Rounding moves a t by ~2^-22 (|origin - o| + q 2^e) / |d|; the boxes are padded by 1e-5 of the mesh
// diagonal for exactly this.  A miss sorts to the end with t = +inf.
template <typename R>
__device__ inline void quant_node_visit(uint4 w0, uint4 w1, uint4 w2, uint4 w3, V3<R> o, V3<R> inv_d, R tmin, R (&tc)[4], uint32_t (&lc)[4])
{
    const R ax = ((R)__uint_as_float(w0.x) - o.x) * inv_d.x, ay = ((R)__uint_as_float(w0.y) - o.y) * inv_d.y,
            az = ((R)__uint_as_float(w0.z) - o.z) * inv_d.z;
    const R bx = (R)__uint_as_float((w0.w & 0xFFu) << 23) * inv_d.x, by = (R)__uint_as_float((w0.w & 0xFF00u) << 15) * inv_d.y,
            bz = (R)__uint_as_float((w0.w & 0xFF0000u) << 7) * inv_d.z;
    const bool ngx = inv_d.x < R(0), ngy = inv_d.y < R(0), ngz = inv_d.z < R(0);
    const uint32_t qnx = ngx ? w2.w : w2.x, qfx = ngx ? w2.x : w2.w;      // near / far plane bytes of the 4 children
    const uint32_t qny = ngy ? w3.x : w2.y, qfy = ngy ? w2.y : w3.x;
    const uint32_t qnz = ngz ? w3.y : w2.z, qfz = ngz ? w2.z : w3.y;
    lc[0] = w1.x; lc[1] = w1.y; lc[2] = w1.z; lc[3] = w1.w;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const R tnx = fma_r((R)((qnx >> (8 * c)) & 0xFFu), bx, ax), tfx = fma_r((R)((qfx >> (8 * c)) & 0xFFu), bx, ax);
        const R tny = fma_r((R)((qny >> (8 * c)) & 0xFFu), by, ay), tfy = fma_r((R)((qfy >> (8 * c)) & 0xFFu), by, ay);
        const R tnz = fma_r((R)((qnz >> (8 * c)) & 0xFFu), bz, az), tfz = fma_r((R)((qfz >> (8 * c)) & 0xFFu), bz, az);
        const R tn = max_r(max_r(tnx, tny), max_r(tnz, R(0)));
        const R tf = min_r(min_r(tfx, tfy), min_r(tfz, tmin));
        tc[c] = (tn <= tf && lc[c] != DRT_BVH_LEAF) ? tn : (R)INFINITY;
    }
}

// near-to-far order of four (t, link) pairs: a 5-comparator network (a miss sorts to the end with t = +inf)
template <typename R>
__device__ inline void sort4_by_t(R (&tc)[4], uint32_t (&lc)[4])
{
#define DRT_CSWAP(i, j) { const bool sw = tc[j] < tc[i]; const R tt = sw ? tc[j] : tc[i]; const R tu = sw ? tc[i] : tc[j]; \
                          const uint32_t lt = sw ? lc[j] : lc[i]; const uint32_t lu = sw ? lc[i] : lc[j];                \
                          tc[i] = tt; tc[j] = tu; lc[i] = lt; lc[j] = lu; }
    DRT_CSWAP(0, 1) DRT_CSWAP(2, 3) DRT_CSWAP(0, 2) DRT_CSWAP(1, 3) DRT_CSWAP(1, 2)
#undef DRT_CSWAP
}

// the triangles of one leaf (<= 4, all requested before the first is tested: one round trip per leaf): closest hit so far
// in (tmin, prim, best_flat); exact ties keep the primitive that comes first in the flattened scene (pathtracer.hpp:80)
template <typename R>
__device__ inline void leaf_visit(const typename Q4<R>::T* __restrict__ tri, uint32_t link, int n_shapes, V3<R> o, V3<R> d,
                                  R& tmin, int& prim, uint32_t& best_flat)
{
    typedef typename Q4<R>::T R4;
    const uint32_t first = (link & 0x7FFFFFFFu) >> 3, count = link & 7u;
    R4 ta[4], tb[4], tcc[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (j < count) {
            const R4* __restrict__ tp = tri + (size_t)(first + j) * 3;
            ta[j] = tp[0];
            tb[j] = tp[1];
            tcc[j] = tp[2];
        }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (j < count) {
            R t;
            if (tri_intersect(mk<R>(ta[j].x, ta[j].y, ta[j].z), mk<R>(ta[j].w, tb[j].x, tb[j].y),
                              mk<R>(tb[j].z, tb[j].w, tcc[j].x), o, d, t)) {
                const uint32_t flat = pid_unpack(tcc[j].z);
                if (t < tmin || (t == tmin && flat < best_flat)) {
                    tmin = t;
                    prim = n_shapes + (int)pid_unpack(tcc[j].y);
                    best_flat = flat;
                }
            }
        }
}

template <typename R>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 ? DRT_WALK_MIN_BLOCKS : 1))
k_intersect_mesh(BatchArgs a, const DevScene<R>* __restrict__ sc, DevBvh<R> bvh, HitRec<R>* hit,
                 const uint32_t* __restrict__ cand, const typename Q4<R>::T* __restrict__ cand_a,
                 const typename Q4<R>::T* __restrict__ cand_b, uint32_t* __restrict__ cand_count, uint32_t cand_cap,
                 uint32_t n_lists, uint32_t group, uint32_t perm_mul, unsigned long long* __restrict__ total)
{
    // (group = lists handed out per pull: 1 for k_intersect's lists, 4 for the shorter per-region lists of k_shade.
    //  perm_mul, coprime to the number of groups: pull number n is group (n * perm_mul) mod n_groups -- consecutive pulls
    //  of a counter land all over the frame.  Handed out in order, counter c's groups are the regions c, c + 64, ...:
    //  with two regions per image row that is the same eight ROWS for every sample, and the counters whose rows cross
    //  the mesh hold most of the work.)
    typedef typename Q4<R>::T R4;
    constexpr uint32_t LDS_NODES = DRT_BVH_LDS_NODES;
    __shared__ uint4 s_node[LDS_NODES][4];
    __shared__ uint32_t s_stack[DRT_BVH_STACK][DRT_BLOCK];
    const uint32_t n_lds = bvh.n_top < LDS_NODES ? bvh.n_top : LDS_NODES;
    for (uint32_t i = threadIdx.x; i < n_lds * 4; i += blockDim.x)
        s_node[i >> 2][i & 3] = bvh.node[i];
    __syncthreads();

    const uint32_t tid = threadIdx.x;
    const int n_shapes = sc->n_shapes;

    // the wave's stream: whole candidate lists (k_intersect), pulled from DRT_PULL_COUNTERS device-wide counters (one
    // returning atomic per list; a single address sustains only ~88 of them per microsecond, which a single counter
    // made the floor of every launch: 0.19 ms); the current list's rays are cand[cur_base + cur_off .. cur_base + cur_cnt)
    uint32_t cur_base = 0, cur_cnt = 0, cur_off = 0, cur_list = 0, group_end = 0;
    const uint32_t n_groups = (n_lists + group - 1) / group;
    bool dry = false;                                           // no group of lists left
    bool home_dry = false;                                      // the wave's own counter has run out
    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t home = (blockIdx.x * (DRT_BLOCK / DRT_WAVE) + threadIdx.x / DRT_WAVE) & (DRT_PULL_COUNTERS - 1);
    uint32_t* const ctr = pull_counters(cand_count, n_lists);

#ifdef DRT_WALK_TIMES
    const uint32_t stat_wave = blockIdx.x * (DRT_BLOCK / DRT_WAVE) + threadIdx.x / DRT_WAVE;
    unsigned long long stat_dry_at = 0;
    if (lane == 0 && stat_wave < 8192)
        g_walk_times[stat_wave][0] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef DRT_BVH_STATS
    uint32_t stat_visits = 0;
    float stat_t[DRT_BVH_STACK], stat_cur_t = 0.f;
    int stat_max_sp = 0;
#endif
    bool active = false;
    uint32_t n_walked = 0;                                      // candidate rays this wave took (statistics: total[5])
    uint32_t slot = 0, cur = DRT_BVH_NONE, best_flat = 0xFFFFFFFFu;
    int sp = 0, prim = -1;
    V3<R> o = mk<R>(R(0), R(0), R(0)), d = o, inv_d = o;
    R tmin = (R)INFINITY;

    for (;;) {
        // ---- refill idle lanes from the stream
        if ((uint32_t)__popcll(__ballot(!active)) >= a.bvh_refill) {
            if ((threadIdx.x & 63) == 0) DRT_STAT(7, 1);          // (stats: refill events)
            bool want = !active;
            for (;;) {
                if (cur_off >= cur_cnt) {
                    if (cur_list + 1 < group_end) {             // the next list of the group pulled last
                        ++cur_list;
                        cur_base = cur_list * cand_cap;
                        cur_cnt = __builtin_amdgcn_readfirstlane(cand_count[cur_list]);
                        cur_off = 0;
                        continue;
                    }
                    if (dry)
                        break;
                    // next group of lists: from the home counter while it lasts, then from whichever counter still has
                    // some (every lane looks at one counter; a lost race just looks again)
                    uint32_t grp = 0xFFFFFFFFu;
                    for (;;) {
                        uint32_t c = home;
                        if (home_dry) {
                            // (a device-scope load: another XCD's L2 must not serve a stale counter -- the loop would never end)
                            const uint32_t seen = __hip_atomic_load(ctr + lane * DRT_PULL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const uint64_t left = __ballot((uint64_t)seen * DRT_PULL_COUNTERS + lane < (uint64_t)n_groups);
                            if (left == 0)
                                break;
                            const uint64_t rot = home ? (left >> home) | (left << (64 - home)) : left;
                            c = (home + (uint32_t)__builtin_ctzll(rot)) & (DRT_PULL_COUNTERS - 1);
                        }
                        uint32_t i = 0;
                        if (lane == 0)
                            i = atomicAdd(ctr + c * DRT_PULL_STRIDE, 1u);
                        i = __builtin_amdgcn_readfirstlane(i);
                        if ((uint64_t)i * DRT_PULL_COUNTERS + c < (uint64_t)n_groups) {
                            grp = i * DRT_PULL_COUNTERS + c;
                            break;
                        }
                        home_dry = true;
                    }
                    if (grp == 0xFFFFFFFFu) {
                        dry = true;
#ifdef DRT_WALK_TIMES
                        stat_dry_at = __builtin_amdgcn_s_memrealtime();
#endif
                        break;
                    }
                    grp = (uint32_t)(((uint64_t)grp * perm_mul) % n_groups);
                    cur_list = grp * group;
                    group_end = cur_list + group < n_lists ? cur_list + group : n_lists;
                    cur_base = cur_list * cand_cap;
                    cur_cnt = __builtin_amdgcn_readfirstlane(cand_count[cur_list]);
                    cur_off = 0;
                    continue;
                }
                const uint64_t wmask = __ballot(want);
                if (wmask == 0)
                    break;
                const uint32_t n_want = (uint32_t)__popcll(wmask), avail = cur_cnt - cur_off;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
                if (want && rank < avail) {
                    // the candidate records are read once: non-temporal, so that they do not push the BVH out of the XCD's L2
                    const size_t at = (size_t)cur_base + cur_off + rank;
                    slot = __builtin_nontemporal_load(cand + at);
                    const R4 ca = nt_load(cand_a + at);
                    const R4 cb = nt_load(cand_b + at);
                    o = mk<R>(ca.x, ca.y, ca.z);
                    d = mk<R>(cb.x, cb.y, cb.z);
                    inv_d = mk<R>(div_r(R(1), d.x), div_r(R(1), d.y), div_r(R(1), d.z));   // (f32: v_rcp; the boxes are padded)
                    tmin = ca.w;                                // closest analytic shape (k_intersect)
                    best_flat = pid_unpack(cb.w);
                    prim = -1;                                  // (a triangle, once one wins)
                    cur = 0;            // root
                    sp = 0;
                    want = false;
                    active = true;
                    DRT_STAT(0, 1);
#ifdef DRT_BVH_STATS
                    stat_visits = 0;
                    stat_cur_t = 0.f;
                    stat_max_sp = 0;
#endif
                }
                cur_off += n_want < avail ? n_want : avail;
                n_walked += n_want < avail ? n_want : avail;
            }
        }
        if (!__any(active))
            break;

        // ---- interior nodes: tight loop, leaves postponed; left as soon as too few lanes still
        // descend (the others already hold a leaf and would only wait)
        for (;;) {
            const bool descending = active && !(cur & DRT_BVH_LEAF);
            const uint64_t dmask = __ballot(descending);
            if (dmask == 0)
                break;
            if ((uint32_t)__popcll(dmask) < a.bvh_descend_min &&
                __ballot(active && (cur & DRT_BVH_LEAF) && cur != DRT_BVH_NONE) != 0)
                break;
            if ((threadIdx.x & 63) == 0) DRT_STAT(5, 1);          // (stats: interior wave-iterations)
            if (!descending)
                continue;
            R tc[4];
            uint32_t lc[4];
            DRT_STAT(cur < n_lds ? 1 : 2, 1);
#ifdef DRT_BVH_STATS
            ++stat_visits;
            if (stat_cur_t > (float)tmin) DRT_STAT(8, 1);
#endif
            uint4 w0, w1, w2, w3;
            if (cur < n_lds) {
                w0 = s_node[cur][0]; w1 = s_node[cur][1]; w2 = s_node[cur][2]; w3 = s_node[cur][3];
            } else {
                const uint4* p = bvh.node + (size_t)cur * 4;
                w0 = p[0]; w1 = p[1]; w2 = p[2]; w3 = p[3];
            }
            quant_node_visit<R>(w0, w1, w2, w3, o, inv_d, tmin, tc, lc);
            sort4_by_t<R>(tc, lc);
            // farthest first onto the stack, nearest becomes current
#ifdef DRT_BVH_STATS
            if (tc[3] < (R)INFINITY) stat_t[sp] = (float)tc[3];
            if (tc[2] < (R)INFINITY) stat_t[sp + (tc[3] < (R)INFINITY)] = (float)tc[2];
            if (tc[1] < (R)INFINITY) stat_t[sp + (tc[3] < (R)INFINITY) + (tc[2] < (R)INFINITY)] = (float)tc[1];
#endif
            if (tc[3] < (R)INFINITY) s_stack[sp++][tid] = lc[3];
            if (tc[2] < (R)INFINITY) s_stack[sp++][tid] = lc[2];
            if (tc[1] < (R)INFINITY) s_stack[sp++][tid] = lc[1];
#ifdef DRT_BVH_STATS
            stat_max_sp = sp > stat_max_sp ? sp : stat_max_sp;
            if (tc[0] < (R)INFINITY) stat_cur_t = (float)tc[0];
            else if (sp > 0) stat_cur_t = stat_t[sp - 1];
#endif
            if (tc[0] < (R)INFINITY)
                cur = lc[0];
            else
                cur = sp > 0 ? s_stack[--sp][tid] : DRT_BVH_NONE;
        }
        // ---- leaves
        if ((threadIdx.x & 63) == 0) DRT_STAT(6, 1);              // (stats: outer wave-iterations)
        if (active && (cur & DRT_BVH_LEAF) && cur != DRT_BVH_NONE) {
            const uint32_t first = (cur & 0x7FFFFFFFu) >> 3, count = cur & 7u;
            DRT_STAT(3, 1);
            DRT_STAT(4, count);
#ifdef DRT_BVH_STATS
            if (stat_cur_t > (float)tmin) DRT_STAT(9, 1);
#endif
            // all triangles of the leaf (<= kMaxLeaf = 4) are requested before the first is tested:
            // one round trip to L2 per leaf instead of one per triangle
            R4 ta[4], tb[4], tcc[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (j < count) {
                    const R4* __restrict__ tp = bvh.tri + (size_t)(first + j) * 3;
                    ta[j] = tp[0];
                    tb[j] = tp[1];
                    tcc[j] = tp[2];
                }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
                if (j < count) {
                    R t;
                    if (tri_intersect(mk<R>(ta[j].x, ta[j].y, ta[j].z), mk<R>(ta[j].w, tb[j].x, tb[j].y),
                                      mk<R>(tb[j].z, tb[j].w, tcc[j].x), o, d, t)) {
                        const uint32_t flat = pid_unpack(tcc[j].z);
                        if (t < tmin || (t == tmin && flat < best_flat)) {
                            tmin = t;
                            prim = n_shapes + (int)pid_unpack(tcc[j].y);
                            best_flat = flat;
                        }
                    }
                }
#ifdef DRT_BVH_STATS
            if (sp > 0) stat_cur_t = stat_t[sp - 1];
#endif
            cur = sp > 0 ? s_stack[--sp][tid] : DRT_BVH_NONE;
        }
        if (active && cur == DRT_BVH_NONE) {
#ifdef DRT_BVH_STATS
            {
                DRT_STAT(10 + (stat_max_sp <= 4 ? 0 : (stat_max_sp <= 8 ? 1 : (stat_max_sp <= 12 ? 2 : (stat_max_sp <= 16 ? 3 : (stat_max_sp <= 24 ? 4 : 5))))), 1);
                const uint32_t v = stat_visits;
                const int bin = v <= 1 ? 0 : (v <= 2 ? 1 : (v <= 4 ? 2 : (v <= 8 ? 3 : (v <= 16 ? 4 : (v <= 32 ? 5 : (v <= 64 ? 6 : 7))))));
                atomicAdd(&g_bvh_hist[bin], 1ull);
                if (prim >= 0) atomicAdd(&g_bvh_hist[8 + bin], 1ull);
                atomicAdd(&g_bvh_hist[16 + bin], (unsigned long long)v);
            }
#endif
            if (prim >= 0) {                                    // a triangle beat the analytic hit k_intersect recorded
                HitRec<R> h;
                h.t = tmin;
                h.prim = prim;
                hit[slot] = h;
            }
            active = false;
        }
    }
    if (total && lane == 0 && n_walked)
        atomicAdd(total + 5, (unsigned long long)n_walked);
#ifdef DRT_WALK_TIMES
    if (lane == 0 && stat_wave < 8192) {
        g_walk_times[stat_wave][1] = stat_dry_at;
        g_walk_times[stat_wave][2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}
