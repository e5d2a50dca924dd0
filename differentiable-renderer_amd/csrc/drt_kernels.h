// drt_kernels.h -- the queue wavefront's kernels K1..K5 and what every kernel of the library shares (batch arguments, the RNG,
// the camera, the analytic closest hit, the BxDF sampler).  K2 on triangles: drt_walk.h; K6 / K7: drt_backward.h; the unbiased
// operator's chain kernels: drt_chain.h; the one-launch kernels: drt_path.h, drt_path_mesh.h.  (gfx950 / wave64)
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
    // (the empty asm keeps this load in a block of its own: merged with the one above into ONE load through a selected
    //  generic pointer -- an LDS address cast to flat -- the compiler's address-space test does not assemble in K6's
    //  wave-at-a-time form, ROCm 7.2 / gfx950: "V_CMP_NE_U32_e32 0, $src_shared_base")
    asm volatile("");
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
#ifdef DRT_USER_SHAPES
    if (SPEC && m.type >= DRT_BXDF_USER) {
        // a BxDF of a caller-defined kind (drt_bxdf_kind_desc): its own sample + evaluate over its two draws, from the source
        // hiprtc compiled into this kernel (drt_prog.h); record = (exponent, norm) of the material
        const R p2[2] = {m.exponent, m.norm};
        user_bxdf<R>(m.type - DRT_BXDF_USER, p2, nrm, d, u01(R(0), r1), u01(R(0), r2), wo, q, bs);
        return;
    }
#endif
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

// one queue a TAIL launch of k_shade appends to: the lanes of one depth, that depth's hit lane and candidate lists
template <typename R>
struct TailQueue {
    typename Q4<R>::T* a;                       // the queue's rays ...
    typename Q2<R>::T* b;
    uint2* id;                                  // ... and their (path, RNG key)
    HitRec<R>* hit;                             // hit lane of the queue's depth (the analytic hit; the walk refines it)
    uint32_t* cand;                             // the walk's candidate lists of that depth, one per region: queue slot,
    typename Q4<R>::T* cand_a;                  // (o.xyz, analytic t),
    typename Q4<R>::T* cand_b;                  // (d.xyz, tie-break index)
    uint32_t* cand_count;                       // records in region w's list
    uint32_t* count;                            // rays in region w's queue: the counts row of that depth
};

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
        // (a second test against the boxes two levels down: measured, not kept -- HISTORY.md 3c)
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
#ifndef DRT_SHADE_MIN_BLOCKS
#define DRT_SHADE_MIN_BLOCKS 1       // blocks per CU the f32 diffuse k_shade is compiled for (1 = the compiler's own choice: 66-89 registers, 5-7 waves per
                                     // SIMD).  Config 4's shade launches, 1024 x 1024 x 8, ms per step: 1.38 as it is, 1.51 forced to six, 1.80 to seven
#endif
template <typename R, bool SPEC, bool FUSED, bool CAM = false, bool TAIL = false>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 && SPEC) ? 4 : (sizeof(R) == 4 ? DRT_SHADE_MIN_BLOCKS : 1))
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
        DevBvh<R> bvh_t, TailQueue<R> tq0, TailQueue<R> tq1, uint32_t* __restrict__ cont_row)
{
    // TAIL (scenes with a mesh; nb = 1 or 2): the launch appends to the queue of depth k + 1 (tq0) the rays that reach the
    // bounds of the mesh -- they wait for the BVH walk -- and, nb = 2, takes the others, whose analytic hit is final, through
    // the vertex of depth k + 1 in registers; what leaves THAT vertex goes to the queue of depth k + 2 (tq1).  A queue is
    // appended to by two launches (k - 1: its second stage, k: its first), so a region's fill level is read when the region
    // is begun; cont_row counts the rays that never saw a queue (they are segments too).
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
        pull_counters(tq0.cand_count, a.n_regions)[threadIdx.x * DRT_PULL_STRIDE] = 0;
    uint32_t cnt;
    uint32_t w = next_region<CAM>(a, counts_k, grid_wave(), n_waves, cnt);
    if (w >= a.n_regions)
        return;
    uint32_t off = 0, running = 0;
    // TAIL: fill levels of the region's queues and candidate lists (depth k + 1: what launch k - 1 left there), rays kept in registers
    uint32_t run0 = 0, crun0 = 0, run1 = 0, crun1 = 0, n_cont = 0, nrun0 = 0, ncrun0 = 0;
    if (TAIL) {
        run0 = tq0.count[w];
        crun0 = tq0.cand_count[w];
    }
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
        if (TAIL && more && nw != w) {
            nrun0 = tq0.count[nw];
            ncrun0 = tq0.cand_count[nw];
        }
        if (CAM && have) {
            const uint32_t i = (w << a.region_shift) + off + lane;      // all alive at depth 0: slot == path
            cur.rid.x = i;
            camera_ray<R>(a, i, cur.ra, cur.rb, cur.rid.y);
        }

        bool alive = false, live = have;
        R4 ra = cur.ra, na;
        R2 rb = cur.rb, nb2;
        HitRec<R> hc = cur.h;                                      // (TAIL: the final hit of the ray in registers)
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
                    h = hc;
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
            if (TAIL) {
                // the ray this vertex produced: its analytic hit while it is in registers; does it reach the bounds of the mesh?
                bool reach = false;
                HitRec<R> hn;
                hn.t = (R)INFINITY;
                hn.prim = -1;
                if (alive) {
                    hn = tail_closest_hit(sc, recs, na, nb2);
                    const V3<R> o2 = mk<R>(na.x, na.y, na.z), d2 = mk<R>(na.w, nb2.x, nb2.y);
                    const V3<R> inv2 = mk<R>(div_r(R(1), d2.x), div_r(R(1), d2.y), div_r(R(1), d2.z));   // (f32: v_rcp; the bounds are padded)
                    R tn;
                    reach = box_hit(mk<R>(bvh_t.lo[0], bvh_t.lo[1], bvh_t.lo[2]), mk<R>(bvh_t.hi[0], bvh_t.hi[1], bvh_t.hi[2]), o2, inv2, hn.t, tn);
                }
                const bool last = it + 1 >= nb;
                const bool enq = alive && (reach || last);             // queued: waits for the walk, or the launch ends here
                const TailQueue<R>& q = it == 0 ? tq0 : tq1;
                const uint32_t qrun = it == 0 ? run0 : run1, qcrun = it == 0 ? crun0 : crun1;
                uint32_t n_enq, n_reach;
                const uint32_t ns = (w << a.region_shift) + qrun + wave_rank(enq, n_enq);
                const uint32_t rk = wave_rank(reach, n_reach);
                if (enq) {
                    q.a[ns] = na;
                    q.b[ns] = nb2;
                    q.id[ns] = cur.rid;
                    q.hit[ns] = hn;
                }
                if (reach) {
                    const size_t at = ((size_t)w << a.region_shift) + qcrun + rk;
                    const uint32_t flat = hn.prim >= 0 ? (uint32_t)sc->flat[hn.prim] : 0xFFFFFFFFu;
                    R4 ca, cb;
                    ca.x = na.x; ca.y = na.y; ca.z = na.z; ca.w = hn.t;
                    cb.x = na.w; cb.y = nb2.x; cb.z = nb2.y; cb.w = pid_pack(R(0), flat);
                    q.cand[at] = ns;
                    q.cand_a[at] = ca;
                    q.cand_b[at] = cb;
                }
                if (it == 0) { run0 += n_enq; crun0 += n_reach; }
                else { run1 += n_enq; crun1 += n_reach; }
                if (last)
                    break;
                const bool go = alive && !reach;
                const uint32_t n_go = (uint32_t)__popcll(__ballot(go));
                if (n_go == 0)
                    break;
                n_cont += n_go;
                ra = na;
                rb = nb2;
                hc = hn;
                live = go;
                continue;
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
        if (!TAIL) {
            uint32_t n_alive;
            const uint32_t ns = (w << a.region_shift) + running + wave_rank(alive, n_alive);
            if (alive) {
                next_a[ns] = na;
                next_b[ns] = nb2;
                next_id[ns] = cur.rid;
            }
            running += n_alive;
        }
        if (nw != w) {                                         // region finished
            if (lane == 0) {
                if (!TAIL && k + nb < a.depth_cap)             // (row depth_cap counts capped paths, see above)
                    counts_next[w] = running;
                if (CAM)
                    counts_k[w] = cnt;                         // depth 0: every path of the region
                if (TAIL) {
                    if (k + 1 < a.depth_cap)
                        tq0.count[w] = run0;
                    tq0.cand_count[w] = crun0;
                    if (nb > 1) {
                        if (k + 2 < a.depth_cap)
                            tq1.count[w] = run1;
                        tq1.cand_count[w] = crun1;
                        if (n_cont)
                            cont_row[w] += n_cont;
                    }
                }
            }
            running = 0;
            run0 = nrun0;
            crun0 = ncrun0;
            run1 = 0;
            crun1 = 0;
            n_cont = 0;
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
