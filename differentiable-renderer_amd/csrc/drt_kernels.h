// drt_kernels.h -- the wavefront kernels K1..K7 (gfx950 / wave64).
//
//   K1 raygen      Camera::sample (camera.hpp:51-60) + depth-0 roulette (pathtracer.hpp:128)
//   K2 intersect   Pathtracer::raycast (pathtracer.hpp:72-89) over Plane/Sphere records
//   K3 shade       Pathtracer::scatter (pathtracer.hpp:91-115): emission, BxDF sample + eval,
//                  throughput update, next-depth roulette, tape write, and the queue append
//                  (K4: wave ballot + prefix into the wave's own queue region, no atomics)
//   K5 film        the per-pixel mean of render.cpp:76-82
//   K6 backward    reverse sweep of the per-bounce tape = the backward functors of
//                  vector.hpp:418-484 in closed form
//   K7 gradreduce  VariableNode::backward's `m_grad += grad` (vector.hpp:185-188), fixed order
//
// Data layout in HBM (R = float, 16-byte lanes; R = double doubles every lane):
//   ray_a[2][N]  (o.x, o.y, o.z, d.x)         dense by queue slot, ping-pong per bounce
//   ray_b[2][N]  (d.y, d.z)                    8 bytes: K2 reads ray_a + ray_b = 24 B, all used
//   ray_id[2][N] (path index, RNG path key)    8 bytes, K3 only
//   hit[N]       (t, shape index | -1)         dense by queue slot
//   tape[D][N]   (m_k, colour param | emission param << 16)   by path index: everything later
//                passes need about vertex k.  T_{k+1} = T_k * colour * m_k and
//                L_k = E_k / p_k + colour_k * m_k * L_{k+1}, so neither the throughput nor the
//                radiance travels through the queue: the tape walk (K6, or k_radiance when only
//                the image is wanted) rebuilds both from 8 bytes per vertex
//   nv[N]        vertices of the path
//   lacc[N]      (L.rgb, -) radiance of the path = L_0, written by the tape walk, read by K5
//   counts[D+1][n_regions]  queue lengths per depth and region (device-resident: no host round
//                trip per bounce)
// Path index i of a batch = (s - s0) * Pb + (pixel - p0): sample-major, so neighbouring lanes
// are neighbouring pixels (coherent rays, coalesced film reads).
//
// Queue regions (K4): the queue of every depth is cut into n_regions regions of region_size
// slots; wave w of the grid owns region w at EVERY depth.  It reads its live rays from the
// front of its region and appends the survivors to the front of the same region of the other
// ping-pong buffer: slot = region base + running count (SGPR) + prefix rank of the lane in the
// wave ballot (v_mbcnt).  No atomics (a single queue-tail word saturates at ~88 returning
// atomics/us on MI355X, which capped the first version of K1/K3 at ~3 ms per launch), the order
// of paths is preserved, and the result is bitwise reproducible.
#pragma once

#include "drt_device.h"
#include "drt_prog.h"

struct BatchArgs {
    // batch geometry
    uint32_t n_paths;        // Pb * Sb
    uint32_t Pb, p0;         // pixels in the batch, first shard-local pixel
    uint32_t Sb, s0;         // samples in the batch, first sample
    uint32_t n_regions, region_size, region_shift;   // queue regions, one wave each; region_size = 1 << region_shift >= 64
    uint32_t rr_threshold;   // r31 < rr_threshold  <=>  double(r31) / RAND_MAX < absorb (exact)
    uint32_t bvh_refill, bvh_descend_min;   // traversal knobs (defaults DRT_BVH_REFILL / DRT_BVH_DESCEND_MIN)
    // image / sharding
    int32_t W, H, spp;
    int32_t shard, n_shards, band;
    // integrator
    int32_t min_bounces, depth_cap;
    int32_t cap_is_roulette;   // the cap is the depth where absorb == 1 kills every path: the reference
                               // still draws its roulette number there (a user max_depth draws nothing)
    int32_t cap_draws;         // the roulette number of the cap's depth counts as drawn (the unbiased operator's draw
                               // bookkeeping): cap_is_roulette, or the cap is the library's own DRT_MAX_DEPTH -- the reference,
                               // which has no cap, draws there, and where that draw ends the path the two stay in step
    double absorb;
    uint32_t seed;
    uint32_t rng_stream;     // drt_rng_stream(seed, 0): the h-seed of every path of the frame (at most 2^32 camera samples)
    // camera (double: per-path work, not per-segment)
    double eye[3], fwd[3], right[3], up[3];
    double tan_half, aspect;
};

// shard-local pixel -> global pixel (y * W + x); rows are dealt to shards in bands
__device__ inline uint32_t global_pixel(const BatchArgs& a, uint32_t lp)
{
    uint32_t ly = lp / (uint32_t)a.W, x = lp - ly * (uint32_t)a.W;
    uint32_t y = ly;
    if (a.n_shards > 1) {
        uint32_t b = ly / (uint32_t)a.band, r = ly - b * (uint32_t)a.band;
        y = (b * (uint32_t)a.n_shards + (uint32_t)a.shard) * (uint32_t)a.band + r;
    }
    return y * (uint32_t)a.W + x;
}

// The n-th draw of the path whose index has the low word `key` (include/drt_hip.h).  The first hash round depends on the
// draw index only: where the lanes of a wave stand at the same index (the lockstep kernels) the compiler evaluates it on
// the scalar unit, and a draw costs the lanes one XOR and one hash round -- what the 32-bit-key scheme of rounds 1-2 cost.
__device__ inline uint32_t rng_draw(uint32_t stream, uint32_t key, uint32_t n)
{
    return drt_rng_combine(drt_rng_index_hash(stream, n), key);
}

// wave vote on a bool without the detour through a vector register (HIP's __ballot / __any take an int: v_cndmask + v_cmp)
__device__ inline uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ inline bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// wave index in the grid, as a scalar
__device__ inline uint32_t grid_wave()
{
    return __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / DRT_WAVE);
}

// wave-local compaction step: rank of this lane among the alive lanes, and their number
__device__ inline uint32_t wave_rank(bool alive, uint32_t& n_alive)
{
    const uint64_t mask = __ballot(alive);
    n_alive = (uint32_t)__popcll(mask);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// uniform in [0,1] from a 31-bit draw, R = float: no fp64 (u and 1-u both from exact integers)
__device__ inline float u01(float, uint32_t r) { return (float)r * (float)(1.0 / DRT_RAND_MAX_D); }
__device__ inline double u01(double, uint32_t r) { return u01_f64(r); }
__device__ inline float one_minus_u01(float, uint32_t r) { return (float)(2147483647u - r) * (float)(1.0 / DRT_RAND_MAX_D); }
__device__ inline double one_minus_u01(double, uint32_t r) { return 1.0 - u01_f64(r); }

// Draw bookkeeping (draw order: SURVEY 3.1).  A path segment that starts at depth s has a BASE =
// the index of its first BxDF draw (theta at depth s; the roulette draw of depth s, if any, comes
// right before it).  The theta draw of depth k >= s is then
//   base + 2 (k - s) + #{roulette draws at depths s+1 .. k} ,
// phi follows it, and the roulette draw of depth k+1 (if k+1 >= min_bounces) follows phi.
// Camera paths: s = 0, base = 2 camera draws (+1 if depth 0 has a roulette draw).
__device__ inline uint32_t draw_offset(int k, int s, int min_bounces)
{
    const int first_rr = min_bounces > s + 1 ? min_bounces : s + 1;
    const int rr = k - first_rr + 1;
    return 2u * (uint32_t)(k - s) + (uint32_t)(rr > 0 ? rr : 0);
}
__device__ inline uint32_t camera_draw_base(int min_bounces) { return 2u + (min_bounces <= 0 ? 1u : 0u); }

// Camera ray of batch-local path i (camera.hpp:51-60, in double like the reference: the pixel
// jitter decides which surface a path starts on) and the RNG key of its path.
template <typename R>
__device__ inline void camera_ray(const BatchArgs& a, uint32_t i, typename Q4<R>::T& ra, typename Q2<R>::T& rb,
                                  uint32_t& key)
{
    const uint32_t sl = i / a.Pb, pl = i - sl * a.Pb;
    const uint32_t gpix = global_pixel(a, a.p0 + pl);
    const uint64_t path = (uint64_t)gpix * (uint64_t)a.spp + (uint64_t)(a.s0 + sl);
    key = (uint32_t)path;
    const uint32_t y = gpix / (uint32_t)a.W, x = gpix - y * (uint32_t)a.W;
    // camera.hpp:53-58
    const double u1 = (double)rng_draw(a.rng_stream, key, 0) / DRT_RAND_MAX_D;
    const double u2 = (double)rng_draw(a.rng_stream, key, 1) / DRT_RAND_MAX_D;
    const double s = ((double)x + u1) / (double)a.W;
    const double t = ((double)y + u2) / (double)a.H;
    const double cs = (2. * s - 1.) * a.aspect * a.tan_half;
    const double ct = (2. * t - 1.) * a.tan_half;
    double dx = a.fwd[0] + cs * a.right[0] - ct * a.up[0];
    double dy = a.fwd[1] + cs * a.right[1] - ct * a.up[1];
    double dz = a.fwd[2] + cs * a.right[2] - ct * a.up[2];
    const double inv = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
    dx *= inv; dy *= inv; dz *= inv;
    ra.x = (R)a.eye[0]; ra.y = (R)a.eye[1]; ra.z = (R)a.eye[2]; ra.w = (R)dx;
    rb.x = (R)dy; rb.y = (R)dz;
}

// ---- K2 ---------------------------------------------------------------------------------------
// Streaming kernel: a 16-byte and an 8-byte load, one 8-byte store per ray (32 B, all used); the shape loop index is
// wave-uniform so the records arrive through the scalar cache into SGPRs.  K2 appends nothing,
// so it does not need the one-wave-per-region mapping: a persistent grid sweeps the 64-slot
// chunks of all regions in address order (neighbouring waves stream neighbouring kilobytes, which
// keeps DRAM pages open) and skips the chunks beyond a region's live count.  Two chunks are in
// flight per wave (their loads are issued before either is consumed).
// Closest hit of NR independent rays in one pass over the shapes: each shape record is fetched
// once (scalar load) for all of them, and their dependency chains interleave.
#ifndef DRT_K2_RAYS
#define DRT_K2_RAYS 2
#endif
template <typename R, int NR>
__device__ inline void closest_hit_n(const DevScene<R>* __restrict__ sc, int n_shapes,
                                     const typename Q4<R>::T (&ra)[NR], const typename Q2<R>::T (&rb)[NR],
                                     HitRec<R> (&h)[NR])
{
    V3<R> o[NR], d[NR];
    R tmin[NR];
    int prim[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        o[r] = mk<R>(ra[r].x, ra[r].y, ra[r].z);
        d[r] = mk<R>(ra[r].w, rb[r].x, rb[r].y);
        tmin[r] = (R)INFINITY;
        prim[r] = -1;
    }
    // The loop is wave-uniform: its control flow and record fetches run on the CU's one scalar unit,
    // which all four SIMDs share (the fused shade kernel issues 0.8 scalar-pipe instructions per vector
    // instruction: tools/pmc_issue.sh).  So it is kept lean in scalar instructions: the type comes
    // from two 64-bit masks held in scalar registers (a bit test, no load), the record is one 16-byte
    // scalar load whose latency the other waves hide (software pipelining the fetch cost seven
    // register moves per shape and was 2-3 % slower), and the plane and the sphere tests sit behind a
    // real branch (the empty asm keeps the compiler from if-converting it back into "compute both,
    // select").
    const unsigned long long planes = sc->plane_mask, spheres = sc->sphere_mask;
    for (int s = 0; s < n_shapes; ++s) {
        const R p0 = sc->shapes[s].p[0], p1 = sc->shapes[s].p[1], p2 = sc->shapes[s].p[2], p3 = sc->shapes[s].p[3];
        if ((planes >> s) & 1ull) {
            const V3<R> n = mk<R>(p0, p1, p2);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const R t = div_r(dot(o[r], n) - p3, -dot(d[r], n));            // shape.hpp:49-59
                if (t > R(0) && !(t >= tmin[r])) {                              // pathtracer.hpp:80
                    tmin[r] = t;
                    prim[r] = s;
                }
            }
        } else if ((spheres >> s) & 1ull) {
            asm volatile("" ::: "memory");
            DevShape<R> sh;
            sh.p[0] = p0; sh.p[1] = p1; sh.p[2] = p2; sh.p[3] = p3;
            sh.type = DRT_SHAPE_SPHERE;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                R t;
                if (shape_intersect(sh, o[r], d[r], t) && !(t >= tmin[r])) {
                    tmin[r] = t;
                    prim[r] = s;
                }
            }
        }                                 // a mesh record: its triangles are k_intersect_mesh's business
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        h[r].t = tmin[r];
        h[r].prim = prim[r];
    }
}

// f32, one ray: the closest hit through the scene's intersection program (DevScene::items): two adjacent
// planes (spheres) are tested together with packed dot products (v_pk_mul/fma/add_f32) -- 18 instead
// of 38 (34 instead of 80) vector instructions per pair, with half the loop control.  Measured 3 % on
// the diffuse and 8 % on the specular shade kernel (packed f32 ops take twice the pipe time of plain
// ones on gfx950, so most of the gain is the scalar side).  Shapes are visited in scene order and the
// first of a pair is accepted first, so ties resolve exactly like the sequential loop
// (pathtracer.hpp:80).
__device__ inline void hit_accept(float t, int s, float& tmin, int& prim)
{
    if (t > 0.f && !(t >= tmin)) {
        tmin = t;
        prim = s;
    }
}

__device__ inline HitRec<float> closest_hit_packed(const DevScene<float>* __restrict__ sc, float4 ra, float2 rb)
{
    const V3<float> o = mk<float>(ra.x, ra.y, ra.z), d = mk<float>(ra.w, rb.x, rb.y);
    const drt_f2 ox = {o.x, o.x}, oy = {o.y, o.y}, oz = {o.z, o.z};
    const drt_f2 dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
    float tmin = INFINITY;
    int prim = -1;
    const int n_items = sc->n_items;
    const unsigned long long pairs = sc->item_pair, spheres = sc->item_sphere, skips = sc->item_skip;
    int s = 0;
    for (int i = 0; i < n_items; ++i) {
        const float* __restrict__ rec = sc->items[i];
        const bool pair = (pairs >> i) & 1ull;
        if ((skips >> i) & 1ull) {
            s += 1;
        } else if (!((spheres >> i) & 1ull)) {
            if (pair) {
                asm volatile("" ::: "memory");
                const drt_f2 nx = {rec[0], rec[1]}, ny = {rec[2], rec[3]}, nz = {rec[4], rec[5]}, off = {rec[6], rec[7]};
                drt_f2 h = ox * nx;
                h = __builtin_elementwise_fma(oy, ny, h);
                h = __builtin_elementwise_fma(oz, nz, h);
                h = h - off;
                drt_f2 den = dx * nx;
                den = __builtin_elementwise_fma(dy, ny, den);
                den = __builtin_elementwise_fma(dz, nz, den);
                hit_accept(h.x * __builtin_amdgcn_rcpf(-den.x), s, tmin, prim);       // shape.hpp:49-59
                hit_accept(h.y * __builtin_amdgcn_rcpf(-den.y), s + 1, tmin, prim);
            } else {
                asm volatile("" ::: "memory");
                const V3<float> n = mk<float>(rec[0], rec[1], rec[2]);
                hit_accept(div_r(dot(o, n) - rec[3], -dot(d, n)), s, tmin, prim);
            }
            s += pair ? 2 : 1;
        } else {
            if (pair) {
                asm volatile("" ::: "memory");
                const drt_f2 cx = {rec[0], rec[1]}, cy = {rec[2], rec[3]}, cz = {rec[4], rec[5]}, rr = {rec[6], rec[7]};
                const drt_f2 ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                drt_f2 bd = ocx * dx;                                         // shape.hpp:78-103, two spheres
                bd = __builtin_elementwise_fma(ocy, dy, bd);
                bd = __builtin_elementwise_fma(ocz, dz, bd);
                drt_f2 cc = ocx * ocx;
                cc = __builtin_elementwise_fma(ocy, ocy, cc);
                cc = __builtin_elementwise_fma(ocz, ocz, cc);
                cc = cc - rr * rr;
                const drt_f2 b = bd * drt_f2{2.f, 2.f};
                const drt_f2 disc = __builtin_elementwise_fma(b, b, cc * drt_f2{-4.f, -4.f});
                const drt_f2 sq = {sqrt_r(disc.x > 0.f ? disc.x : 0.f), sqrt_r(disc.y > 0.f ? disc.y : 0.f)};
                const drt_f2 t1 = (-b - sq) * drt_f2{0.5f, 0.5f}, t2 = (sq - b) * drt_f2{0.5f, 0.5f};
                const float ta = t1.x > 0.f ? t1.x : t2.x, tb = t1.y > 0.f ? t1.y : t2.y;
                if (disc.x >= 0.f) hit_accept(ta, s, tmin, prim);
                if (disc.y >= 0.f) hit_accept(tb, s + 1, tmin, prim);
            } else {
                asm volatile("" ::: "memory");
                DevShape<float> sh;
                sh.p[0] = rec[0]; sh.p[1] = rec[1]; sh.p[2] = rec[2]; sh.p[3] = rec[3];
                sh.type = DRT_SHAPE_SPHERE;
                float t;
                if (shape_intersect(sh, o, d, t))
                    hit_accept(t, s, tmin, prim);
            }
            s += pair ? 2 : 1;
        }
    }
    HitRec<float> h;
    h.t = tmin;
    h.prim = prim;
    return h;
}

__device__ inline HitRec<double> closest_hit_packed(const DevScene<double>* __restrict__ sc, double4 ra, double2 rb)
{
    const double4 ra1[1] = {ra};
    const double2 rb1[1] = {rb};
    HitRec<double> h1[1];
    closest_hit_n<double, 1>(sc, sc->n_shapes, ra1, rb1, h1);      // the f64 verification mode keeps the literal loop
    return h1[0];
}

// slot of this lane in chunk c, or 0xFFFFFFFF when the lane has no live ray there
__device__ inline uint32_t chunk_slot(const BatchArgs& a, const uint32_t* __restrict__ counts_k,
                                      uint32_t c, uint32_t n_chunks, uint32_t lane)
{
    if (c >= n_chunks)
        return 0xFFFFFFFFu;
    const uint32_t cpr_shift = a.region_shift - 6;
    const uint32_t w = c >> cpr_shift;
    const uint32_t off = (c - (w << cpr_shift)) * DRT_WAVE + lane;
    const uint32_t cnt = __builtin_amdgcn_readfirstlane(counts_k[w]);
    return off < cnt ? (w << a.region_shift) + off : 0xFFFFFFFFu;
}

// the walk's pull counters live behind the n_lists list lengths, on a 128-byte boundary
__device__ inline uint32_t* pull_counters(uint32_t* cand_count, uint32_t n_lists)
{
    return cand_count + ((n_lists + DRT_PULL_STRIDE - 1) / DRT_PULL_STRIDE) * DRT_PULL_STRIDE;
}

// K2 as a kernel of its own: the textbook wavefront (DRT_RENDER_UNFUSED; analytic scenes -- the fused routes fold it into
// K3, and in scenes with a mesh every ray is intersected with the analytic shapes by the launch that PRODUCES it, which
// also builds the BVH walk's candidate lists: tail_emit below).  Persistent grid, DRT_K2_RAYS rays per lane in flight.
template <typename R>
__global__ void __launch_bounds__(DRT_BLOCK)
k_intersect(BatchArgs a, const DevScene<R>* __restrict__ sc, const typename Q4<R>::T* __restrict__ ray_a,
            const typename Q2<R>::T* __restrict__ ray_b, HitRec<R>* __restrict__ hit,
            const uint32_t* __restrict__ counts_k, unsigned long long* __restrict__ total)
{
    typedef typename Q4<R>::T R4;
    typedef typename Q2<R>::T R2;
    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t n_waves = gridDim.x * (DRT_BLOCK / DRT_WAVE);
    const uint32_t n_chunks = a.n_regions << (a.region_shift - 6);
    const int n_shapes = sc->n_shapes;
    constexpr int NR = DRT_K2_RAYS;
    const uint32_t gw = grid_wave();
    uint32_t n_rays = 0;                                        // rays this wave intersected (statistics: total[4])
    for (uint32_t c = gw; c < n_chunks; c += NR * n_waves) {
        uint32_t slot[NR];
        R4 ra[NR];
        R2 rb[NR];
        HitRec<R> h[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            slot[r] = chunk_slot(a, counts_k, c + (uint32_t)r * n_waves, n_chunks, lane);
            ra[r] = R4{};
            rb[r] = R2{};
            if (slot[r] != 0xFFFFFFFFu) { ra[r] = ray_a[slot[r]]; rb[r] = ray_b[slot[r]]; }
            n_rays += (uint32_t)__popcll(__ballot(slot[r] != 0xFFFFFFFFu));
        }
        closest_hit_n<R, NR>(sc, n_shapes, ra, rb, h);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (slot[r] != 0xFFFFFFFFu)
                hit[slot[r]] = h[r];
    }
    if (total && lane == 0 && n_rays)
        atomicAdd(total + 4, (unsigned long long)n_rays);       // (integer: order-independent)
}

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

// ---- K3 ---------------------------------------------------------------------------------------
template <typename R>
struct SceneLds {
    DevScene<R> sc;
    R params[DRT_LDS_PARAMS * 3];
};

// Copy the USED part of the scene (header, n_shapes shapes, n_materials materials, n_emitters
// emitter ids, up to DRT_LDS_PARAMS parameters) into LDS: ~0.5 KB for the Cornell box.
template <typename R>
__device__ inline void stage_scene(SceneLds<R>& lds, const DevScene<R>* __restrict__ sc,
                                   const R* __restrict__ params)
{
    const int ns = sc->n_shapes, nm = sc->n_materials, ne = sc->n_emitters;
    if (threadIdx.x < 4)
        reinterpret_cast<int*>(&lds.sc)[threadIdx.x] = reinterpret_cast<const int*>(sc)[threadIdx.x];
    {
        const int* src = reinterpret_cast<const int*>(sc->shapes);
        int* dst = reinterpret_cast<int*>(lds.sc.shapes);
        for (int i = threadIdx.x; i < ns * (int)(sizeof(DevShape<R>) / sizeof(int)); i += blockDim.x)
            dst[i] = src[i];
    }
    {
        const int* src = reinterpret_cast<const int*>(sc->materials);
        int* dst = reinterpret_cast<int*>(lds.sc.materials);
        for (int i = threadIdx.x; i < nm * (int)(sizeof(DevMaterial<R>) / sizeof(int)); i += blockDim.x)
            dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < ne; i += blockDim.x)
        lds.sc.emitter_param[i] = sc->emitter_param[i];
    const int np = sc->n_params < DRT_LDS_PARAMS ? sc->n_params : DRT_LDS_PARAMS;
    for (int i = threadIdx.x; i < np * 3; i += blockDim.x)
        lds.params[i] = params[i];
    __syncthreads();
}

template <typename R, bool ALL_LDS = false>
__device__ inline V3<R> load_param(const SceneLds<R>& lds, const R* __restrict__ params, int id)
{
    if (ALL_LDS || id < DRT_LDS_PARAMS)
        return mk<R>(lds.params[id * 3], lds.params[id * 3 + 1], lds.params[id * 3 + 2]);
    return mk<R>(params[id * 3], params[id * 3 + 1], params[id * 3 + 2]);
}

// what K3 consumes per ray: its queue lanes and the hit record
template <typename R>
struct ShadeIn {
    typename Q4<R>::T ra;
    typename Q2<R>::T rb;
    uint2 rid;
    HitRec<R> h;
};

template <typename R>
__device__ inline void load_shade_in(ShadeIn<R>& in, uint32_t slot, bool have, bool with_hit,
                                     const typename Q4<R>::T* __restrict__ ray_a,
                                     const typename Q2<R>::T* __restrict__ ray_b,
                                     const uint2* __restrict__ ray_id,
                                     const HitRec<R>* __restrict__ hit)
{
    if (have) {
        in.ra = ray_a[slot];
        in.rb = ray_b[slot];
        in.rid = ray_id[slot];
        if (with_hit)
            in.h = hit[slot];
    }
}

// what was hit: position, normal (as the reference's Shape::normal returns it), material, emitter, and the colour parameter
// of its BxDF (the material's, or -- a mesh face with drt_mesh_desc::face_param -- the face's own)
template <typename R, bool MESHES = true>
__device__ inline void resolve_hit(const SceneLds<R>& lds, const typename Q4<R>::T* __restrict__ tri_shade,
                                   int prim, V3<R> P, V3<R>& nrm, int& material, int& emitter, uint32_t& cparam)
{
    if (!MESHES || prim < lds.sc.n_shapes) {
        const DevShape<R>& sh = lds.sc.shapes[prim];
        nrm = shape_normal(sh, P);
        material = sh.material;
        emitter = sh.emitter;
        cparam = material >= 0 ? (uint32_t)lds.sc.materials[material].param : DRT_ID_NONE;
    } else {                                       // triangle: per-triangle record
        const typename Q4<R>::T ts = tri_shade[prim - lds.sc.n_shapes];
        const uint32_t ids = pid_unpack(ts.w);
        nrm = mk<R>(ts.x, ts.y, ts.z);
        cparam = ids & 0xFFFFu;
        material = ((ids >> 16) & 0xFFu) == 0xFFu ? -1 : (int)((ids >> 16) & 0xFFu);
        emitter = (ids >> 24) == 0xFFu ? -1 : (int)(ids >> 24);
    }
}

// BxDF::sample + BxDF::operator() for one vertex: r1, r2 are the two 31-bit draws; returns the
// sampled direction wo, its pdf q and the scalar bs with f = colour * bs.
//   Diffuse  (bxdf.hpp:56-83):  theta = asin(sqrt(u1)) => sin = sqrt(u1), cos = sqrt(1-u1); 1 - u1
//            comes from the exact integer RAND_MAX - r so cos (and the pdf) is never rounded to 0
//   Specular (bxdf.hpp:85-124): cos^2(theta) = u1^(2/(e+2)); sin^2 formed in double
template <typename R, bool SPEC>
__device__ inline void sample_bxdf(const DevMaterial<R>& m, V3<R> nrm, V3<R> d, uint32_t r1, uint32_t r2,
                                   V3<R>& wo, R& q, R& bs)
{
    // The two end points of the theta draw are singular: u1 = 1 gives cos(theta) = 0 and pdf 0 in
    // the diffuse sampler (0/0 here; the reference survives only because cos(asin(1.0)) is 6e-17 in
    // fp64), u1 = 0 or 1 give pdf 0 in the specular one (inf/NaN in the reference too).  At ~1e9
    // draws per render a 2^-31 event happens, so the draw is kept one step inside the interval.
    if (SPEC && m.type == DRT_BXDF_MIRROR) {
        // bxdf.hpp:126-144 (repaired): dir = reflect(dir_in, n), pdf 1, f = 1 / cos on every channel;
        // its two draws are skipped, not used.  The colour parameter is the scene's internal
        // constant (1, 1, 1), so tape, radiance and gradient kernels need no special case.
        const V3<R> wi = -d;
        wo = reflect(wi, nrm);
        q = R(1);
        bs = R(1) / dot(nrm, wo);
        return;
    }
    r1 = min(max(r1, 1u), 2147483646u);                      // (v_med3_u32)
    R sphi, cphi;
    sincos_2pi_u31(r2, &sphi, &cphi);                          // phi = 2 pi u2
    V3<R> tg, bt;
    make_frame(nrm, tg, bt);
    if (!SPEC || m.type == DRT_BXDF_DIFFUSE) {
        const R st = sqrt_r(u01(R(0), r1)), ct = sqrt_r(one_minus_u01(R(0), r1));
        wo = tg * (cphi * st) + bt * (sphi * st) + nrm * ct;
        q = ct * (R)(1.0 / DRT_PI);
        bs = (R)(1.0 / DRT_PI);                                // bxdf.hpp:63-67: color / pi
    } else {
        // cos^2 = u^(2/(e+2)) and sin^2 = 1 - cos^2.  f64 forms them literally.  f32 goes through
        // x = log(u) * 2/(e+2) <= 0: cos^2 = exp(x), sin^2 = -expm1(x), with log(u) taken as
        // log1p(-(1-u)) from the EXACT integer RAND_MAX - r1 when u is near 1 -- both keep full
        // relative accuracy where the literal form cancels, without a double-precision pow.
        R ct, st, x_half = R(0);
        if (sizeof(R) == 4) {
            // (drt_sincos.h: hardware exp2 / log2 and short series instead of libm's expf / logf / log1pf / expm1f)
            const float x = drt_log_u31(r1) * (2.0f / ((float)m.exponent + 2.0f));
            ct = (R)sqrt_r(drt_exp_nonpos(x));
            st = (R)sqrt_r(drt_one_minus_exp(x));
            x_half = (R)(0.5f * x);
        } else {
            const double c2 = pow((double)r1 / DRT_RAND_MAX_D, 2.0 / ((double)m.exponent + 2.0));
            ct = sqrt_r((R)c2);
            st = sqrt_r((R)(1.0 - c2));
        }
        const V3<R> wi = -d;
        V3<R> hv = tg * (cphi * st) + bt * (sphi * st) + nrm * ct;
        if (dot(hv, wi) < R(0))
            hv = reflect(hv, nrm);
        wo = reflect(wi, hv);
        // pdf: cos^(e+1) = exp((e+1) * log(cos)), and log(cos) = x / 2 is already known in f32
        q = m.norm * (sizeof(R) == 4 ? (R)drt_exp_nonpos((float)((m.exponent + R(1)) * x_half)) : pow_r(ct, m.exponent + R(1))) * st;
        // bxdf.hpp:91-104 re-derives the half vector as normalize(dir_in + dir_out).  That sum is
        // 2 (h . wi) h: when h is nearly perpendicular to wi it cancels, and in f32 it can cancel
        // to exactly 0 (-> NaN; seen once per ~3e7 paths at depth 12).  f32 therefore uses the
        // identity halfway = sign(h . wi) h; f64 keeps the literal form (exact parity, and enough
        // headroom).
        R ch;
        if (sizeof(R) == 4) {
            ch = dot(hv, wi) < R(0) ? -dot(nrm, hv) : dot(nrm, hv);
        } else {
            const V3<R> hw = normalize(wi + wo);
            ch = dot(nrm, hw);
        }
        R s2 = (R(1) - ch) * (R(1) + ch);
        s2 = s2 > R(0) ? s2 : R(0);
        bs = m.norm * pow_weight_r(ch, m.exponent) * sqrt_r(s2);
    }
}

// first region >= w (stepping by n_waves) that has live rays; its count in cnt
__device__ inline uint32_t next_live_region(const uint32_t* __restrict__ counts_k, uint32_t w, uint32_t n_waves,
                                            uint32_t n_regions, uint32_t& cnt)
{
    cnt = 0;
    while (w < n_regions) {
        cnt = __builtin_amdgcn_readfirstlane(counts_k[w]);
        if (cnt)
            break;
        w += n_waves;
    }
    return w;
}

// ---- K6 ---------------------------------------------------------------------------------------
// One thread per path.  The tape holds 8 bytes per vertex (m_k, parameter ids); the prefix
// throughputs T_k are rebuilt in registers with the very expression K3 used
// (T_{k+1} = T_k * colour * m_k), DRT_TAPE_CHUNK vertices at a time, then the chunk is walked
// deepest-first with the suffix radiance in registers:
//   L_k = E_k / p_k + colour_k * m_k * L_{k+1}
//   d/dE_k      += g * T_k / p_k
//   d/dcolour_k += g * T_k * m_k * L_{k+1}
// (closed form of the backward functors vector.hpp:418-484).  Paths longer than one
// chunk rebuild the prefix product of the earlier chunks from the tape again.
// Parameter ids < DRT_FAST_PARAMS accumulate in registers (compare-select, no atomics, fixed
// order => bitwise reproducible); other ids use fp64 atomics on the gradient vector.
#define DRT_TAPE_CHUNK 8

// Gradient accumulators of one thread.
//   NP = 4 or 8 (the scene has at most NP parameters): NP x 3 registers, conditional adds with a
//        compile-time parameter index -- no memory traffic, no waits, fixed order.
//   NP = 0 (general): a column per thread in LDS, acc[row = param * 3 + channel][thread], for ids
//        < DRT_FAST_PARAMS (bank = thread % 32: conflict-free; plain read/add/write -- LDS float
//        ATOMICS were measured 4x slower than the rest of the kernel) and fp64 global atomics for
//        the others.  The read-add-write chains serialise on lgkmcnt, so NP > 0 is ~2x faster.
template <typename R, int NP>
struct GradAcc {
    R r[NP][3];
    __device__ inline void init(R (*)[DRT_BLOCK])
    {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            r[p][0] = r[p][1] = r[p][2] = R(0);
    }
    __device__ inline void add(R (*)[DRT_BLOCK], double* __restrict__, uint32_t id, V3<R> v)
    {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool sel = id == (uint32_t)p;
            r[p][0] += sel ? v.x : R(0);
            r[p][1] += sel ? v.y : R(0);
            r[p][2] += sel ? v.z : R(0);
        }
    }
    __device__ inline double get(R (*)[DRT_BLOCK], int row) const
    {
        double v = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (row == p * 3 + c)
                    v = (double)r[p][c];
        return v;
    }
};

// f32, parameters in registers: one-hot weights and packed FMAs (v_pk_fma_f32) -- xy of every
// parameter as one pair, the z components of two parameters as another: 14 VALU per add for NP = 4
// instead of a compare + three selects + three adds per parameter.
template <int NP>
struct GradAccF32 {
    static_assert(NP % 2 == 0, "z components are paired");
    drt_f2 xy[NP], zz[NP / 2];
    __device__ inline void init(float (*)[DRT_BLOCK])
    {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            xy[p] = drt_f2{0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NP / 2; ++q)
            zz[q] = drt_f2{0.f, 0.f};
    }
    __device__ inline void add(float (*)[DRT_BLOCK], double* __restrict__, uint32_t id, V3<float> v)
    {
        float w[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
            w[p] = id == (uint32_t)p ? 1.f : 0.f;
        const drt_f2 vxy = drt_f2{v.x, v.y}, vzz = drt_f2{v.z, v.z};
#pragma unroll
        for (int p = 0; p < NP; ++p)
            xy[p] = __builtin_elementwise_fma(drt_f2{w[p], w[p]}, vxy, xy[p]);
#pragma unroll
        for (int q = 0; q < NP / 2; ++q)
            zz[q] = __builtin_elementwise_fma(drt_f2{w[2 * q], w[2 * q + 1]}, vzz, zz[q]);
    }
    __device__ inline double get(float (*)[DRT_BLOCK], int row) const
    {
        double v = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (row == p * 3 + 0) v = (double)xy[p].x;
            if (row == p * 3 + 1) v = (double)xy[p].y;
            if (row == p * 3 + 2) v = (double)((p & 1) ? zz[p / 2].y : zz[p / 2].x);
        }
        return v;
    }
};
template <> struct GradAcc<float, 4> : GradAccF32<4> {};
template <> struct GradAcc<float, 8> : GradAccF32<8> {};

// General case (any number of parameters): ONE set of fp64 accumulators per block in LDS, row = param * 3 + channel
// for the first DRT_LDS_PARAMS parameters, updated with LDS atomics (ds_add_f64; lanes that add to the same row
// serialise inside the LDS, which costs ~5x the register path but involves no other CU).  The first version used
// fp64 GLOBAL atomics on the gradient vector for ids >= 8: every thread of the chip adding to the same few
// addresses -- 117 ms instead of 0.4 for a mesh with seven per-face albedos.  Parameters beyond DRT_LDS_PARAMS (none
// in practice: a scene has at most 64 materials and 64 emitters) still go to the gradient vector directly.
// The pointer travels through the accumulator interface as R (*)[DRT_BLOCK]; it points at DRT_LDS_PARAMS * 3 doubles.
template <typename R>
struct GradAcc<R, 0> {
    // The first DRT_FAST_PARAMS parameters stay in registers here too: in a room with a mesh they are the walls' colours and
    // the light -- most vertices of most paths -- and as LDS atomics they all land on the same few words (config 4 with an
    // albedo per face, 50,884 parameters: K6 4.1 ms that way).  One-hot accumulation like GradAcc<R, 8>; ids beyond add nothing there.
    GradAcc<R, DRT_FAST_PARAMS> fast;
    __device__ inline void init(R (*acc)[DRT_BLOCK])
    {
        fast.init(acc);
        double* blk = reinterpret_cast<double*>(acc);
        for (int r = threadIdx.x; r < DRT_LDS_PARAMS * 3; r += DRT_BLOCK)
            blk[r] = 0.0;                        // (visible to the block after stage_scene's barrier)
    }
    __device__ inline void add(R (*acc)[DRT_BLOCK], double* __restrict__ grad, uint32_t id, V3<R> v)
    {
        fast.add(acc, grad, id, v);
        if (id >= DRT_FAST_PARAMS) {
            double* dst = id < DRT_LDS_PARAMS ? reinterpret_cast<double*>(acc) + id * 3 : grad + id * 3;
            atomicAdd(dst + 0, (double)v.x);
            atomicAdd(dst + 1, (double)v.y);
            atomicAdd(dst + 2, (double)v.z);
        }
    }
    __device__ inline double get(R (*acc)[DRT_BLOCK], int row) const { return fast.get(acc, row); }
};

// end of a gradient kernel: this block's sums -> gpart[block][row_stride] (fixed-order reduction over blocks: K7)
template <typename R, int NP>
__device__ inline void flush_grad_block(GradAcc<R, NP>& ga, R (*acc)[DRT_BLOCK], double (*red)[DRT_FAST_PARAMS * 3],
                                        double* __restrict__ gpart, int n_rows, int row_stride)
{
    if (NP == 0) {
        // the register rows (ids < DRT_FAST_PARAMS): thread -> wave by shuffles, then one LDS add per wave and row
        const int lane0 = threadIdx.x & (DRT_WAVE - 1);
#pragma unroll
        for (int r = 0; r < DRT_FAST_PARAMS * 3; ++r) {
            double v = ga.get(acc, r);
#pragma unroll
            for (int off = DRT_WAVE / 2; off > 0; off >>= 1)
                v += __shfl_down(v, off);
            if (lane0 == 0 && v != 0.0)
                atomicAdd(reinterpret_cast<double*>(acc) + r, v);
        }
        __syncthreads();
        const double* blk = reinterpret_cast<const double*>(acc);
        for (int r = threadIdx.x; r < n_rows; r += DRT_BLOCK)
            gpart[(size_t)blockIdx.x * row_stride + r] = blk[r];
        return;
    }
    // thread registers -> wave (shuffles) -> block (LDS), fp64, fixed order
    const int lane = threadIdx.x & (DRT_WAVE - 1), wave = threadIdx.x / DRT_WAVE;
#pragma unroll
    for (int r = 0; r < (NP > 0 ? NP * 3 : 1); ++r) {
        double v = ga.get(acc, r);
#pragma unroll
        for (int off = DRT_WAVE / 2; off > 0; off >>= 1)
            v += __shfl_down(v, off);
        if (lane == 0)
            red[wave][r] = v;
    }
    __syncthreads();
    if (threadIdx.x < DRT_FAST_PARAMS * 3) {
        double v = 0;
        if ((int)threadIdx.x < NP * 3)
            for (int w = 0; w < DRT_BLOCK / DRT_WAVE; ++w)
                v += red[w][threadIdx.x];
        gpart[(size_t)blockIdx.x * row_stride + threadIdx.x] = v;
    }
}

// The walk of ONE path's tape (see the K6 comment above); every gradient contribution is handed
// to acc.add(acc_lds, grad, parameter id, value).
template <typename R, bool SMALL, typename Acc>
__device__ inline V3<R> backward_path(const BatchArgs& a, const SceneLds<R>& lds, const R* __restrict__ params,
                                     const TapeRec<R>* __restrict__ tape, size_t N, uint32_t i, int K, V3<R> g,
                                     R inv_p_rr, Acc& acc, R (*acc_lds)[DRT_BLOCK], double* __restrict__ grad,
                                     const TapeRec<R>* first_chunk = nullptr)
{
    V3<R> Ln = mk<R>(R(0), R(0), R(0));
    for (int c0 = ((K - 1) / DRT_TAPE_CHUNK) * DRT_TAPE_CHUNK; c0 >= 0; c0 -= DRT_TAPE_CHUNK) {
        // prefix throughput at the start of this chunk (only for paths longer than a chunk)
        V3<R> T = mk<R>(R(1), R(1), R(1));
        for (int j = 0; j < c0; ++j) {
            const TapeRec<R> tr = tape[(size_t)j * N + i];
            T = T * load_param<R, SMALL>(lds, params, (int)(tr.ids & 0xFFFFu)) * tr.m;
        }
        R Tx[DRT_TAPE_CHUNK], Ty[DRT_TAPE_CHUNK], Tz[DRT_TAPE_CHUNK], M[DRT_TAPE_CHUNK];
        uint32_t ID[DRT_TAPE_CHUNK];
        TapeRec<R> trs[DRT_TAPE_CHUNK];
        if (first_chunk && c0 == 0) {
            // vertices 0..7 were requested together with the path's vertex count (k_backward)
#pragma unroll
            for (int j = 0; j < DRT_TAPE_CHUNK; ++j)
                trs[j] = first_chunk[j];
        } else {
#pragma unroll
            for (int j = 0; j < DRT_TAPE_CHUNK; ++j)
                if (c0 + j < K)
                    trs[j] = tape[(size_t)(c0 + j) * N + i];   // independent loads, all in flight
        }
#pragma unroll
        for (int j = 0; j < DRT_TAPE_CHUNK; ++j) {
            if (c0 + j < K) {
                ID[j] = trs[j].ids;
                M[j] = trs[j].m;
                Tx[j] = T.x; Ty[j] = T.y; Tz[j] = T.z;
                const uint32_t cid = ID[j] & 0xFFFFu;
                if (cid != DRT_ID_NONE)
                    T = T * load_param<R, SMALL>(lds, params, (int)cid) * M[j];
            }
        }
#pragma unroll
        for (int j = DRT_TAPE_CHUNK - 1; j >= 0; --j) {
            const int k = c0 + j;
            if (k < K) {
                const uint32_t cid = ID[j] & 0xFFFFu, eid = ID[j] >> 16;
                const R inv_pk = k >= a.min_bounces ? inv_p_rr : R(1);
                const V3<R> adj = g * mk<R>(Tx[j], Ty[j], Tz[j]);
                V3<R> Lk = mk<R>(R(0), R(0), R(0));
                if (eid != DRT_ID_NONE) {
                    acc.add(acc_lds, grad, eid, adj * inv_pk);
                    Lk = load_param<R, SMALL>(lds, params, (int)eid) * inv_pk;
                }
                if (cid != DRT_ID_NONE) {
                    const V3<R> wgt = Ln * M[j];
                    acc.add(acc_lds, grad, cid, adj * wgt);
                    Lk = Lk + load_param<R, SMALL>(lds, params, (int)cid) * wgt;
                }
                Ln = Lk;
            }
        }
    }
    return Ln;          // L_0: the radiance of the path
}

// the seed a path is back-propagated with: (1, 1, 1) (render.cpp:80), the caller's per-pixel adjoint, or -- DRT_RENDER_LOSS_L2,
// `radiance` given -- the derivative of the per-sample squared error against the target image, 2 (L_path - target_pixel)
// (README.md:93-98: loss = loss_func(radiance); loss.backward())
template <typename R>
__device__ inline V3<R> path_seed(const BatchArgs& a, const float* __restrict__ adjoint, uint32_t i,
                                  const typename Q4<R>::T* __restrict__ radiance = nullptr)
{
    if (!adjoint)
        return mk<R>(R(1), R(1), R(1));                       // render.cpp:80
    const uint32_t gp = global_pixel(a, a.p0 + i % a.Pb);
    const V3<R> t = mk<R>((R)adjoint[(size_t)gp * 3], (R)adjoint[(size_t)gp * 3 + 1], (R)adjoint[(size_t)gp * 3 + 2]);
    if (radiance) {
        const typename Q4<R>::T L = radiance[i];
        return mk<R>(R(2) * (L.x - t.x), R(2) * (L.y - t.y), R(2) * (L.z - t.z));
    }
    return t;
}

// next region >= w this wave shades; CAM: depth 0, region w holds its share of the batch's paths
template <bool CAM>
__device__ inline uint32_t next_region(const BatchArgs& a, const uint32_t* __restrict__ counts_k, uint32_t w,
                                       uint32_t n_waves, uint32_t& cnt)
{
    if (!CAM)
        return next_live_region(counts_k, w, n_waves, a.n_regions, cnt);
    const uint32_t begin = w << a.region_shift;
    cnt = w < a.n_regions ? (a.n_paths - begin < a.region_size ? a.n_paths - begin : a.region_size) : 0u;
    return w;
}

// Persistent blocks; every wave walks the regions w, w + n_waves, ... it owns in this launch and
// shades them chunk by chunk.  The loads of the NEXT chunk (same region or the next live one) are
// issued before the current chunk is shaded, so a wave always has one chunk of loads in flight.
// SPEC = false instantiations carry no specular code (and fewer registers) for all-diffuse scenes.
// FUSED = true: the closest hit over the analytic shapes is computed HERE from the ray just loaded
// (K2 folded into K3): no hit lane, no second read of the ray -- 72 instead of 120 bytes per
// segment.  Used whenever nothing else needs the hit records (no mesh, no unbiased chain vertices).
// CAM (fused launches that start at depth 0, every path alive there): the camera ray is generated in
// place -- K1 folded in too: nothing is read from the queue, the row of depth 0 is written here.
// (A launch that takes every path from the eye to its end needs neither queue nor tape: that is k_path, drt_path.h.)
// the analytic closest hit of k_shade's tail: the LDS program in f32 (no scalar loop: the launch stays bandwidth-bound),
// the literal loop in the f64 verification mode
__device__ inline HitRec<float> tail_closest_hit(const DevScene<float>* __restrict__ sc, const ProgRecs<0>& recs, float4 ra, float2 rb)
{
    return closest_hit_prog<SigNone>(sc, recs, mk<float>(ra.x, ra.y, ra.z), mk<float>(ra.w, rb.x, rb.y));
}
__device__ inline HitRec<double> tail_closest_hit(const DevScene<double>* __restrict__ sc, const ProgRecs<0>&, double4 ra, double2 rb)
{
    return closest_hit_packed(sc, ra, rb);
}

// ---- the TAIL step, shared by the kernels that PRODUCE rays in scenes with a mesh (k_raygen, k_shade, k_adj_vertex) ------
// The ray a lane has just written to queue slot `slot` of region w is intersected with the analytic shapes while it is still
// in registers (the hit lane of the ray's depth gets the result) and, if it reaches the bounds of the mesh before that hit,
// its complete record -- slot, origin, direction, analytic t, tie-break index: 36 bytes -- is appended to the REGION's
// candidate list (wave ballot + prefix rank, no atomics) for the BVH walk.  No kernel ever re-reads a ray just to find out
// whether the walk must see it.
__device__ inline void stage_tail_program(ProgLds& s_prog, const DevScene<float>* __restrict__ scf)
{
    if (threadIdx.x < DRT_PROG_SORTED_MAX) {
        s_prog.rec[threadIdx.x] = *reinterpret_cast<const float4*>(scf->sorted[threadIdx.x]);
        s_prog.shape[threadIdx.x] = scf->sorted_shape[threadIdx.x];
    }
    if (threadIdx.x < 8)
        s_prog.kind_begin[threadIdx.x] = scf->kind_begin[threadIdx.x];
}
__device__ inline void stage_tail_program(ProgLds&, const DevScene<double>* __restrict__) { }   // (f64: the literal loop)

template <typename R>
__device__ inline void tail_emit(const BatchArgs& a, const DevScene<R>* __restrict__ sc, const ProgRecs<0>& recs, const DevBvh<R>& bvh_t,
                                 bool alive, uint32_t slot, typename Q4<R>::T na, typename Q2<R>::T nb, uint32_t w, uint32_t& cand_running,
                                 HitRec<R>* __restrict__ hit_next, uint32_t* __restrict__ cand, typename Q4<R>::T* __restrict__ cand_a,
                                 typename Q4<R>::T* __restrict__ cand_b)
{
    typedef typename Q4<R>::T R4;
    bool reach = false;
    HitRec<R> hn;
    hn.t = (R)INFINITY;
    hn.prim = -1;
    if (alive) {
        hn = tail_closest_hit(sc, recs, na, nb);
        hit_next[slot] = hn;
        const V3<R> o2 = mk<R>(na.x, na.y, na.z), d2 = mk<R>(na.w, nb.x, nb.y);
        const V3<R> inv2 = mk<R>(div_r(R(1), d2.x), div_r(R(1), d2.y), div_r(R(1), d2.z));   // (f32: v_rcp; the bounds are padded)
        R tn;
        reach = box_hit(mk<R>(bvh_t.lo[0], bvh_t.lo[1], bvh_t.lo[2]), mk<R>(bvh_t.hi[0], bvh_t.hi[1], bvh_t.hi[2]), o2, inv2, hn.t, tn);
        // (Measured and not kept: a second test against the <= 16 boxes two levels down -- 21 % of the rays that reach the
        //  bounds of the 50,880-triangle sphere die within two levels -- took 5 % of the candidates' walk time away,
        //  4.14 -> 4.06 ms per step, and added 0.48 ms to the shade launches that run it.)
    }
    uint32_t n_reach;
    const uint32_t rk = wave_rank(reach, n_reach);
    if (reach) {
        const size_t at = ((size_t)w << a.region_shift) + cand_running + rk;
        const uint32_t flat = hn.prim >= 0 ? (uint32_t)sc->flat[hn.prim] : 0xFFFFFFFFu;
        R4 ca, cb;
        ca.x = na.x; ca.y = na.y; ca.z = na.z; ca.w = hn.t;
        cb.x = na.w; cb.y = nb.x; cb.z = nb.y; cb.w = pid_pack(R(0), flat);
        cand[at] = slot;
        cand_a[at] = ca;
        cand_b[at] = cb;
    }
    cand_running += n_reach;
}

// ---- K1 ---------------------------------------------------------------------------------------
// One wave per queue region: generates the camera rays of its region's paths and compacts the
// ones that survive the depth-0 roulette to the front of the region.  TAIL (scenes with a mesh): plus the TAIL step above --
// K1 and K2's analytic pass in one launch: the camera rays are never read back (32 B written and 24 B read per path saved).
template <typename R, bool TAIL = false>
__global__ void __launch_bounds__(DRT_BLOCK)
k_raygen(BatchArgs a, const DevScene<R>* __restrict__ sc, typename Q4<R>::T* __restrict__ ray_a, typename Q2<R>::T* __restrict__ ray_b,
         uint2* __restrict__ ray_id, uint32_t* __restrict__ nv, uint32_t* __restrict__ counts,
         DevBvh<R> bvh_t, HitRec<R>* __restrict__ hit_next, uint32_t* __restrict__ cand,
         typename Q4<R>::T* __restrict__ cand_a, typename Q4<R>::T* __restrict__ cand_b, uint32_t* __restrict__ cand_count)
{
    typedef typename Q4<R>::T R4;
    __shared__ ProgLds s_prog;
    ProgRecs<0> recs;
    recs.lds = &s_prog;
    if (TAIL) {
        stage_tail_program(s_prog, sc);
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < DRT_PULL_COUNTERS)     // the walk's list counters (it runs after this kernel)
            pull_counters(cand_count, a.n_regions)[threadIdx.x * DRT_PULL_STRIDE] = 0;
    }
    const uint32_t w = grid_wave();
    if (w >= a.n_regions)
        return;
    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t begin = w * a.region_size;
    const uint32_t end = min(begin + a.region_size, a.n_paths);
    uint32_t running = 0, cand_running = 0;
    for (uint32_t off = begin; off < end; off += DRT_WAVE) {
        const uint32_t i = off + lane;
        bool alive = i < end;
        R4 ra;
        typename Q2<R>::T rb;
        uint2 rid;
        if (alive) {
            uint32_t key;
            camera_ray<R>(a, i, ra, rb, key);
            // pathtracer.hpp:128 at depth 0
            if (a.depth_cap <= 0)
                alive = false;
            else if (a.min_bounces <= 0 && rng_draw(a.rng_stream, key, 2) < a.rr_threshold)
                alive = false;
            rid.x = i; rid.y = key;
            if (!alive)
                nv[i] = 0;
        }
        uint32_t n_alive;
        const uint32_t slot = begin + running + wave_rank(alive, n_alive);
        if (alive) {
            ray_a[slot] = ra;
            ray_b[slot] = rb;
            ray_id[slot] = rid;
        }
        if (TAIL)
            tail_emit<R>(a, sc, recs, bvh_t, alive, slot, ra, rb, w, cand_running, hit_next, cand, cand_a, cand_b);
        running += n_alive;
    }
    if (lane == 0) {
        counts[w] = running;
        if (TAIL)
            cand_count[w] = cand_running;
    }
}

// TAIL (scenes with a mesh, one bounce per launch): the ray this launch PRODUCES is intersected with the analytic shapes
// right here, while it is still in registers, and handed to the BVH walk if it reaches the mesh -- K2's analytic pass
// (k_intersect) then only runs for the camera rays, and the queue is not read a second time (24 bytes per ray).  The
// candidate list of a region is the region's own span of the candidate arrays; hit_next is the hit lane of the NEXT depth.
template <typename R, bool SPEC, bool FUSED, bool CAM = false, bool TAIL = false>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 && SPEC) ? 4 : 1)
k_shade(BatchArgs a, int k, int nb, const DevScene<R>* __restrict__ sc, const R* __restrict__ params,
        const typename Q4<R>::T* __restrict__ ray_a, const typename Q2<R>::T* __restrict__ ray_b,
        const uint2* __restrict__ ray_id, const HitRec<R>* __restrict__ hit,
        typename Q4<R>::T* __restrict__ next_a, typename Q2<R>::T* __restrict__ next_b,
        uint2* __restrict__ next_id,
        TapeRec<R>* __restrict__ tape_k, uint32_t* __restrict__ nv,
        uint32_t* __restrict__ counts_k, uint32_t count_stride,
        const typename Q4<R>::T* __restrict__ tri_shade,
        int seg_start, const uint32_t* __restrict__ draw_base,
        typename Q4<R>::T* __restrict__ save_a, typename Q2<R>::T* __restrict__ save_b,
        HitRec<R>* __restrict__ save_hit,
        DevBvh<R> bvh_t, HitRec<R>* __restrict__ hit_next, uint32_t* __restrict__ cand,
        typename Q4<R>::T* __restrict__ cand_a, typename Q4<R>::T* __restrict__ cand_b, uint32_t* __restrict__ cand_count)
{
    // nb > 1 (FUSED only): the launch takes every ray through nb bounces -- depths k .. k+nb-1 -- in
    // registers; only the survivors of the LAST one are compacted and written back.  Lanes whose
    // path ended in between idle (a few per cent per bounce), in exchange the 64 bytes of queue
    // traffic per ray are paid once per nb segments.  counts_k + j * count_stride is the row of
    // depth k + j: the rows in between are kept up to date with one non-returning atomic per chunk
    // (a region is owned by one wave), the row of depth k + nb is written like before.
    typedef typename Q4<R>::T R4;
    typedef typename Q2<R>::T R2;
    __shared__ SceneLds<R> lds;
    __shared__ ProgLds s_prog;                  // TAIL, f32: the kind-sorted intersection program of the analytic shapes
    ProgRecs<0> recs;
    recs.lds = &s_prog;
    if (TAIL)
        stage_tail_program(s_prog, sc);
    stage_scene(lds, sc, params);               // (ends with a barrier)

    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t n_waves = gridDim.x * (DRT_BLOCK / DRT_WAVE);
    const size_t N = a.n_paths;
    uint32_t* __restrict__ counts_next = counts_k + (size_t)nb * count_stride;

    if (TAIL && blockIdx.x == 0 && threadIdx.x < DRT_PULL_COUNTERS)     // the walk's list counters (it runs after this kernel)
        pull_counters(cand_count, a.n_regions)[threadIdx.x * DRT_PULL_STRIDE] = 0;
    uint32_t cnt;
    uint32_t w = next_region<CAM>(a, counts_k, grid_wave(), n_waves, cnt);
    if (w >= a.n_regions)
        return;
    uint32_t off = 0, running = 0, cand_running = 0;
    ShadeIn<R> cur, nxt;
    bool have = lane < cnt;
    if (!CAM)
        load_shade_in(cur, (w << a.region_shift) + lane, have, !FUSED, ray_a, ray_b, ray_id, hit);

    for (;;) {
        // where the next chunk is, and its loads
        uint32_t nw = w, noff = off + DRT_WAVE, ncnt = cnt;
        if (noff >= cnt) {
            noff = 0;
            nw = next_region<CAM>(a, counts_k, w + n_waves, n_waves, ncnt);
        }
        const bool more = nw < a.n_regions;
        const bool nhave = more && noff + lane < ncnt;
        if (!CAM)
            load_shade_in(nxt, (nw << a.region_shift) + noff + lane, nhave, !FUSED, ray_a, ray_b, ray_id, hit);
        if (CAM && have) {
            const uint32_t i = (w << a.region_shift) + off + lane;      // all alive at depth 0: slot == path
            cur.rid.x = i;
            camera_ray<R>(a, i, cur.ra, cur.rb, cur.rid.y);
        }

        bool alive = false, live = have;
        R4 ra = cur.ra, na;
        R2 rb = cur.rb, nb2;
        const uint32_t pid = cur.rid.x, key = cur.rid.y;
        for (int it = 0;; ++it) {
            const int kk = k + it;
            const R pk = kk >= a.min_bounces ? (R)(1.0 - a.absorb) : R(1);   // pathtracer.hpp:130
            // camera paths start at depth 0 with a closed-form base; re-sampled suffixes (unbiased
            // backward) start at seg_start with a per-path base
            const uint32_t n_off = draw_offset(kk, seg_start, a.min_bounces) + (draw_base ? 0u : camera_draw_base(a.min_bounces));
            const bool next_rr = (kk + 1) >= a.min_bounces;
            const bool next_cap = (kk + 1) >= a.depth_cap;
            TapeRec<R>* __restrict__ tape_kk = tape_k + (size_t)it * N;
            alive = false;
            bool capped = false;                                   // cut short by max_depth (not by the roulette)
            if (live) {
                HitRec<R> h;
                if (FUSED) {
                    h = closest_hit_packed(sc, ra, rb);
                } else {
                    h = cur.h;
                }
                if (save_a && it == 0) {   // unbiased backward: this ray and its hit are the path's next chain vertex
                    save_a[pid] = ra;
                    save_b[pid] = rb;
                    save_hit[pid] = h;
                }
                // what this vertex writes: at most one tape record and, when the path ends here, its length
                TapeRec<R> tr;
                bool write_tape = false, ended = true;
                uint32_t n_vertices = (uint32_t)kk;                // miss: pathtracer.hpp:135
                if (h.prim >= 0) {
                    const V3<R> o = mk<R>(ra.x, ra.y, ra.z);
                    const V3<R> d = mk<R>(ra.w, rb.x, rb.y);
                    const V3<R> P = o + d * h.t;                   // pathtracer.hpp:83
                    V3<R> nrm;
                    int material, emitter;
                    uint32_t cparam;
                    resolve_hit<R, !FUSED>(lds, tri_shade, h.prim, P, nrm, material, emitter, cparam);
                    // emission (pathtracer.hpp:113-114) is only RECORDED here: the tape walk adds it
                    const uint32_t eid = emitter >= 0 ? (uint32_t)lds.sc.emitter_param[emitter] : DRT_ID_NONE;
                    write_tape = true;
                    n_vertices = (uint32_t)kk + 1u;
                    // no BxDF: f = 0 (pathtracer.hpp:38-39); the reference's zero-direction
                    // continuation contributes exactly 0, the path ends here
                    tr.m = R(0);
                    tr.ids = DRT_ID_NONE | (eid << 16);
                    if (material >= 0) {
                        const DevMaterial<R>& m = lds.sc.materials[material];
                        const uint32_t n_theta = n_off + (draw_base ? draw_base[pid] : 0u);
                        V3<R> wo;
                        R q, bs;
                        sample_bxdf<R, SPEC>(m, nrm, d, rng_draw(a.rng_stream, key, n_theta), rng_draw(a.rng_stream, key, n_theta + 1), wo, q, bs);
                        const R c = dot(nrm, wo);                  // pathtracer.hpp:103
                        const R mk_ = bs * c / (q * pk);           // T_{k+1} = T_k * color * m_k
#ifdef DRT_DEBUG_NAN
                        if (!(mk_ == mk_) || mk_ > R(1e30) || mk_ < R(-1e30))
                            printf("[k_shade] pid %u k %d type %d: bs %g c %g q %g pk %g | nrm %g %g %g | d %g %g %g | wo %g %g %g | t %g prim %d\n",
                                   pid, kk, m.type, (double)bs, (double)c, (double)q, (double)pk, (double)nrm.x, (double)nrm.y,
                                   (double)nrm.z, (double)d.x, (double)d.y, (double)d.z, (double)wo.x, (double)wo.y, (double)wo.z,
                                   (double)h.t, h.prim);
#endif
                        // roulette / cap of depth kk+1, decided here so dead rays are never queued
                        alive = !next_cap;
                        if (alive && next_rr)
                            alive = !(rng_draw(a.rng_stream, key, n_theta + 2) < a.rr_threshold);
                        // a user max_depth ends the path here: had the reference's roulette let it live?
                        if (next_cap && !a.cap_is_roulette)
                            capped = !next_rr || !(rng_draw(a.rng_stream, key, n_theta + 2) < a.rr_threshold);
                        tr.m = mk_;
                        tr.ids = cparam | (eid << 16);
                        ended = !alive;
                        const V3<R> no = P + wo * R(1e-3);         // pathtracer.hpp:99
                        na.x = no.x; na.y = no.y; na.z = no.z; na.w = wo.x;
                        nb2.x = wo.y; nb2.y = wo.z;
                    }
                }
                if (write_tape)
                    tape_kk[pid] = tr;
                if (ended)
                    nv[pid] = n_vertices;
            }
            if (next_cap && !a.cap_is_roulette) {                  // row D of the counts: paths the cap cut short
                const uint32_t n_cap = (uint32_t)__popcll(__ballot(capped));
                if (lane == 0 && n_cap)
                    atomicAdd(counts_k + (size_t)(it + 1) * count_stride + w, n_cap);
            }
            if (it + 1 >= nb)
                break;
            // survivors go straight into the next bounce; the row of the depth in between only counts them
            const uint32_t n_mid = (uint32_t)__popcll(__ballot(alive));
            if (n_mid == 0)
                break;                                             // (alive is false in every lane)
            if (lane == 0)
                atomicAdd(counts_k + (size_t)(it + 1) * count_stride + w, n_mid);
            ra = na;
            rb = nb2;
            live = alive;
        }
        uint32_t n_alive;
        const uint32_t ns = (w << a.region_shift) + running + wave_rank(alive, n_alive);
        if (alive) {
            next_a[ns] = na;
            next_b[ns] = nb2;
            next_id[ns] = cur.rid;
        }
        if (TAIL)
            tail_emit<R>(a, sc, recs, bvh_t, alive, ns, na, nb2, w, cand_running, hit_next, cand, cand_a, cand_b);
        running += n_alive;
        if (nw != w) {                                         // region finished
            if (lane == 0) {
                if (k + nb < a.depth_cap)                      // (row depth_cap counts capped paths, see above)
                    counts_next[w] = running;
                if (CAM)
                    counts_k[w] = cnt;                         // depth 0: every path of the region
                if (TAIL)
                    cand_count[w] = cand_running;
            }
            running = 0;
            cand_running = 0;
        }
        if (!more)
            break;
        cur = nxt;
        have = nhave;
        w = nw;
        off = noff;
        cnt = ncnt;
    }
}

// segments of one batch = rays queued at depths 0..D-1, summed over regions -> 64-bit total
__global__ void __launch_bounds__(DRT_BLOCK)
k_sum_counts(const uint32_t* __restrict__ counts, uint32_t n_words, unsigned long long* __restrict__ total,
             uint32_t row_words, unsigned long long read_rows, unsigned long long written_rows, uint32_t cap_row)
{
    // total[0] += all words of rows != cap_row (= segments); total[1] += the rows a shade launch STARTED from (rays
    // read from the queue), total[2] += the rows a launch ended on (survivors written back); row r = bit r of the
    // masks; total[3] += row cap_row (paths that were still alive when the depth cap cut them: never queued)
    __shared__ unsigned long long red[4][DRT_BLOCK / DRT_WAVE];
    unsigned long long v = 0, vr = 0, vw = 0, vc = 0;
    for (uint32_t i = blockIdx.x * DRT_BLOCK + threadIdx.x; i < n_words; i += gridDim.x * DRT_BLOCK) {
        const unsigned long long c = counts[i];
        const uint32_t row = row_words ? i / row_words : 0u;
        if (row == cap_row) { vc += c; continue; }
        v += c;
        if (row < 64u && ((read_rows >> row) & 1ull)) vr += c;
        if (row < 64u && ((written_rows >> row) & 1ull)) vw += c;
    }
    for (int off = DRT_WAVE / 2; off > 0; off >>= 1) {
        v += __shfl_down(v, off);
        vr += __shfl_down(vr, off);
        vw += __shfl_down(vw, off);
        vc += __shfl_down(vc, off);
    }
    if ((threadIdx.x & (DRT_WAVE - 1)) == 0) {
        red[0][threadIdx.x / DRT_WAVE] = v;
        red[1][threadIdx.x / DRT_WAVE] = vr;
        red[2][threadIdx.x / DRT_WAVE] = vw;
        red[3][threadIdx.x / DRT_WAVE] = vc;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long t = 0;
        for (int w = 0; w < DRT_BLOCK / DRT_WAVE; ++w)
            t += red[threadIdx.x][w];
        if (t)
            atomicAdd(total + threadIdx.x, t);   // integer: order-independent
    }
}

// ---- K5 ---------------------------------------------------------------------------------------
template <typename R>
__global__ void __launch_bounds__(DRT_BLOCK)
k_film(BatchArgs a, const typename Q4<R>::T* __restrict__ lacc, double* __restrict__ film)
{
    typedef typename Q4<R>::T R4;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < a.Pb; j += stride) {
        double r = 0, g = 0, b = 0;
        for (uint32_t s = 0; s < a.Sb; ++s) {
            const R4 L = lacc[(size_t)s * a.Pb + j];
            r += (double)L.x; g += (double)L.y; b += (double)L.z;
        }
        double* f = film + (size_t)(a.p0 + j) * 3;
        f[0] += r; f[1] += g; f[2] += b;
    }
}

// film (sums, shard-local) -> out_rgb (means, global row-major float)
__global__ void __launch_bounds__(DRT_BLOCK)
k_resolve(BatchArgs a, uint32_t n_pixels, const double* __restrict__ film, float* __restrict__ out)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    const double inv = 1.0 / (double)a.spp;
    for (uint32_t lp = blockIdx.x * blockDim.x + threadIdx.x; lp < n_pixels; lp += stride) {
        const uint32_t gp = global_pixel(a, lp);
        drt_f3 px;                                   // (one 12-byte store per lane: `out` may be pinned host memory)
        px.x = (float)(film[(size_t)lp * 3 + 0] * inv);
        px.y = (float)(film[(size_t)lp * 3 + 1] * inv);
        px.z = (float)(film[(size_t)lp * 3 + 2] * inv);
        *reinterpret_cast<drt_f3_u*>(out + (size_t)gp * 3) = px;
    }
}

// ---- K6 (kernel; the tape walk and the accumulators it uses are defined above) ----
template <typename R, int NP>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 && NP == 4) ? 4 : 1)
k_backward(BatchArgs a, const DevScene<R>* __restrict__ sc, const R* __restrict__ params,
           const TapeRec<R>* __restrict__ tape, const uint32_t* __restrict__ nv,
           const float* __restrict__ adjoint, double* __restrict__ gpart, double* __restrict__ grad,
           typename Q4<R>::T* __restrict__ lacc, int n_rows, int row_stride,
           const typename Q4<R>::T* radiance_in = nullptr)
{
    // (radiance_in: DRT_RENDER_LOSS_L2 -- the radiance of every path, written by k_radiance before this launch; may alias lacc)
    typedef typename Q4<R>::T R4;
    constexpr bool SMALL = NP > 0;
    __shared__ SceneLds<R> lds;
    __shared__ double acc_d[NP > 0 ? 1 : DRT_LDS_PARAMS * 3];          // NP == 0: the block's accumulators (GradAcc<R, 0>)
    R (*acc)[DRT_BLOCK] = reinterpret_cast<R(*)[DRT_BLOCK]>(acc_d);
    __shared__ double red[DRT_BLOCK / DRT_WAVE][DRT_FAST_PARAMS * 3];
    GradAcc<R, NP> ga;
    ga.init(acc);
    stage_scene(lds, sc, params);

    const size_t N = a.n_paths;
    const R inv_p_rr = (R)(1.0 / (1.0 - a.absorb));
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_paths; i += stride) {
        // The first chunk of the tape is requested WITH the vertex count, not after it: one round trip
        // to memory per path instead of two (records beyond the path's end are read and ignored; the
        // rows exist for every depth below the cap).  Also prefetching the NEXT path's chunk was
        // measured slower: 174 VGPRs, 2 waves per SIMD.
        const int K = (int)nv[i];
        TapeRec<R> first[DRT_TAPE_CHUNK];
#pragma unroll
        for (int j = 0; j < DRT_TAPE_CHUNK; ++j)
            if (j < a.depth_cap)
                first[j] = tape[(size_t)j * N + i];
        V3<R> L0 = mk<R>(R(0), R(0), R(0));
        if (K > 0)
            L0 = backward_path<R, SMALL>(a, lds, params, tape, N, i, K, path_seed<R>(a, adjoint, i, radiance_in), inv_p_rr, ga, acc, grad, first);
        if (lacc) {
            R4 o;
            o.x = L0.x; o.y = L0.y; o.z = L0.z; o.w = R(0);
            lacc[i] = o;
        }
    }

    flush_grad_block<R, NP>(ga, acc, red, gpart, n_rows, row_stride);
}

// Forward-only renders: the radiance of every path from its tape, deepest vertex first --
//   L_k = E_k / p_k + colour_k * m_k * L_{k+1}
// which is the order in which the reference's recursion returns (pathtracer.hpp:104,114,133).
template <typename R>
__global__ void __launch_bounds__(DRT_BLOCK)
k_radiance(BatchArgs a, const DevScene<R>* __restrict__ sc, const R* __restrict__ params,
           const TapeRec<R>* __restrict__ tape, const uint32_t* __restrict__ nv,
           typename Q4<R>::T* __restrict__ lacc)
{
    typedef typename Q4<R>::T R4;
    __shared__ SceneLds<R> lds;
    stage_scene(lds, sc, params);
    const size_t N = a.n_paths;
    const R inv_p_rr = (R)(1.0 / (1.0 - a.absorb));
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_paths; i += stride) {
        const int K = (int)nv[i];
        V3<R> L = mk<R>(R(0), R(0), R(0));
        for (int c0 = ((K - 1) / DRT_TAPE_CHUNK) * DRT_TAPE_CHUNK; c0 >= 0 && K > 0; c0 -= DRT_TAPE_CHUNK) {
            TapeRec<R> trs[DRT_TAPE_CHUNK];
#pragma unroll
            for (int j = 0; j < DRT_TAPE_CHUNK; ++j)
                if (c0 + j < K)
                    trs[j] = tape[(size_t)(c0 + j) * N + i];
#pragma unroll
            for (int j = DRT_TAPE_CHUNK - 1; j >= 0; --j) {
                const int k = c0 + j;
                if (k < K) {
                    const uint32_t cid = trs[j].ids & 0xFFFFu, eid = trs[j].ids >> 16;
                    const R inv_pk = k >= a.min_bounces ? inv_p_rr : R(1);
                    V3<R> Lk = mk<R>(R(0), R(0), R(0));
                    if (eid != DRT_ID_NONE)
                        Lk = load_param(lds, params, (int)eid) * inv_pk;
                    if (cid != DRT_ID_NONE)
                        Lk = Lk + load_param(lds, params, (int)cid) * (L * trs[j].m);
                    L = Lk;
                }
            }
        }
        R4 o;
        o.x = L.x; o.y = L.y; o.z = L.z; o.w = R(0);
        lacc[i] = o;
    }
}

// Gradient-image variant (README.md:142-145 of the reference): the gradient of ONE parameter, kept
// per path instead of reduced -- written to a lacc-shaped buffer that K5 then averages per pixel.
template <typename R>
struct OneParamAcc {
    uint32_t param;
    V3<R> sum;
    __device__ inline void add(R (*)[DRT_BLOCK], double* __restrict__, uint32_t id, V3<R> v)
    {
        if (id == param)
            sum = sum + v;
    }
};

template <typename R>
__global__ void __launch_bounds__(DRT_BLOCK)
k_backward_image(BatchArgs a, const DevScene<R>* __restrict__ sc, const R* __restrict__ params,
                 const TapeRec<R>* __restrict__ tape, const uint32_t* __restrict__ nv,
                 const float* __restrict__ adjoint, uint32_t param, typename Q4<R>::T* __restrict__ gpath,
                 typename Q4<R>::T* __restrict__ lacc)
{
    typedef typename Q4<R>::T R4;
    __shared__ SceneLds<R> lds;
    stage_scene(lds, sc, params);
    const size_t N = a.n_paths;
    const R inv_p_rr = (R)(1.0 / (1.0 - a.absorb));
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n_paths; i += stride) {
        OneParamAcc<R> acc;
        acc.param = param;
        acc.sum = mk<R>(R(0), R(0), R(0));
        const int K = (int)nv[i];
        V3<R> L0 = mk<R>(R(0), R(0), R(0));
        if (K > 0)
            L0 = backward_path<R, false>(a, lds, params, tape, N, i, K, path_seed<R>(a, adjoint, i), inv_p_rr, acc,
                                         (R(*)[DRT_BLOCK]) nullptr, nullptr);
        R4 o;
        o.x = acc.sum.x; o.y = acc.sum.y; o.z = acc.sum.z; o.w = R(0);
        gpath[i] = o;
        if (lacc) {
            o.x = L0.x; o.y = L0.y; o.z = L0.z;
            lacc[i] = o;
        }
    }
}

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

// ---- K7 ---------------------------------------------------------------------------------------
// grad[p] += sum over blocks of gpart[block][p] in a fixed order (deterministic); one block per row p
__global__ void __launch_bounds__(DRT_BLOCK)
k_gradreduce(const double* __restrict__ gpart, int n_blocks, int n_rows, double* __restrict__ grad, int row_stride)
{
    __shared__ double red[DRT_BLOCK];
    const int p = blockIdx.x;
    if (p >= n_rows)
        return;
    double v = 0;
    for (int b = threadIdx.x; b < n_blocks; b += DRT_BLOCK)
        v += gpart[(size_t)b * row_stride + p];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int off = DRT_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        grad[p] += red[0];
}
