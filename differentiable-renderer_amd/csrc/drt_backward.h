// drt_backward.h -- K6 / K7 of the queue wavefront: the reverse sweep of the per-vertex tape (the backward functors of
// vector.hpp:418-484 in closed form), the per-thread / per-block gradient accumulators, the forward-only
// radiance walk, the per-pixel gradient image, and the fixed-order reduction that is VariableNode::backward's
// `m_grad += grad` (vector.hpp:185-188).
#pragma once

#include "drt_kernels.h"

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
// (the general accumulator -- any number of parameters, a wave adds at once: GradAcc<R, 0> -- walks chunks of 4: its adds hold
//  ~60 registers of their own, and at 8 vertices per chunk the kernel runs two waves per SIMD instead of three: config 4 with an
//  albedo per face 3.13 -> 3.01 ms; the register accumulators lose 20 % at 4: their kernel stays at 8)
#ifndef DRT_TAPE_CHUNK_WAVE
#define DRT_TAPE_CHUNK_WAVE 4
#endif

// Gradient accumulators of one thread.
//   NP = 4 or 8 (the scene has at most NP parameters): NP x 3 registers, conditional adds with a
//        compile-time parameter index -- no memory traffic, no waits, fixed order.
//   NP = 0 (general): a column per thread in LDS, acc[row = param * 3 + channel][thread], for ids
//        < DRT_FAST_PARAMS (bank = thread % 32: conflict-free; plain read/add/write -- LDS float
//        ATOMICS were measured 4x slower than the rest of the kernel) and fp64 global atomics for
//        the others.  The read-add-write chains serialise on lgkmcnt, so NP > 0 is ~2x faster.
template <typename R, int NP>
struct GradAcc {
    static constexpr bool kWave = false;     // (add() may be called by any subset of a wave's lanes)
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
    __device__ inline void add_wave(R (*a)[DRT_BLOCK], double* __restrict__ g, uint32_t id, V3<R> v) { add(a, g, id, v); }
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
    static constexpr bool kWave = false;
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
    __device__ inline void add_wave(float (*a)[DRT_BLOCK], double* __restrict__ g, uint32_t id, V3<float> v) { add(a, g, id, v); }
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
// the value lane `byte_address / 4` holds (ds_bpermute)
__device__ inline float lane_read(int byte_address, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_address, __float_as_int(v))); }
__device__ inline double lane_read(int byte_address, double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_address, (int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_address, (int)(uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <typename R>
struct GradAcc<R, 0> {
    static constexpr bool kWave = true;      // K6 calls add_wave() with ALL lanes of the wave (ids of DRT_ID_NONE add nothing)
    // The first DRT_FAST_PARAMS parameters stay in registers (one-hot accumulation like GradAcc<R, 8>; other ids add nothing
    // there): in a room with a mesh they are the walls' colours and the light -- most vertices of most paths -- and as LDS
    // atomics they all land on the same few words (config 4 with an albedo per face: K6 3.9 ms that way, 3.2 ms this way).
    GradAcc<R, DRT_FAST_PARAMS> fast;
    __device__ inline void init(R (*acc)[DRT_BLOCK])
    {
        fast.init(acc);
        double* blk = reinterpret_cast<double*>(acc);
        for (int r = threadIdx.x; r < DRT_LDS_PARAMS * 3; r += DRT_BLOCK)
            blk[r] = 0.0;                        // (visible to the block after stage_scene's barrier)
    }
    // any subset of a wave's lanes (the unbiased operator's chain kernel): per-lane atomics, LDS rows or the gradient vector
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
    // The same for a whole wave at once (every lane calls; id == DRT_ID_NONE: nothing to add).  Parameters beyond the LDS rows
    // go to the gradient vector with fp64 atomics, which execute at the memory side as 64-byte requests: three adds per lane
    // (x, y, z) are three requests of 8 useful bytes each -- 0.16 TB/s of added bytes, an eighth of the chip's atomic rate
    // (config 4 with an albedo per face: K6 4.1 ms).  So the lanes are TRANSPOSED first: in round r lane L carries component
    // (64 r + L) % 3 of the vertex of lane (64 r + L) / 3 -- three neighbouring lanes add the x, y, z of one parameter's row,
    // 24 contiguous bytes, one request (two where the row straddles a line).
    __device__ inline void add_wave(R (*acc)[DRT_BLOCK], double* __restrict__ grad, uint32_t id, V3<R> v)
    {
        fast.add(acc, grad, id, v);
        const bool none = id == DRT_ID_NONE;
        if (!none && id >= DRT_FAST_PARAMS && id < DRT_LDS_PARAMS) {
            // (the block's rows in LDS, addressed as LDS: the low word of a generic LDS pointer is the LDS offset)
            typedef __attribute__((address_space(3))) double lds_double;
            lds_double* dst = (lds_double*)(uint32_t)(uintptr_t)(reinterpret_cast<double*>(acc) + id * 3);
            __hip_atomic_fetch_add(dst + 0, (double)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 1, (double)v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 2, (double)v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const uint32_t fid = (!none && id >= DRT_LDS_PARAMS) ? id : DRT_ID_NONE;
        if (__builtin_amdgcn_ballot_w64(fid != DRT_ID_NONE) == 0)
            return;
        const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
#pragma unroll
        for (uint32_t r = 0; r < 3; ++r) {
            const uint32_t q = r * DRT_WAVE + lane, src = q / 3u, comp = q - src * 3u;
            const int at = (int)(src << 2);                       // ds_bpermute addresses lanes in bytes
            const uint32_t sid = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)fid);
            const R sx = lane_read(at, v.x), sy = lane_read(at, v.y), sz = lane_read(at, v.z);
            const R val = comp == 0 ? sx : (comp == 1 ? sy : sz);
            if (sid != DRT_ID_NONE)
                atomicAdd(grad + (size_t)sid * 3 + comp, (double)val);
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
    constexpr int CH = Acc::kWave ? DRT_TAPE_CHUNK_WAVE : DRT_TAPE_CHUNK;      // vertices per chunk
    V3<R> Ln = mk<R>(R(0), R(0), R(0));
    // (an accumulator that adds a whole wave at a time -- Acc::kWave, the general one -- is called by ALL lanes at every step:
    //  the chunk loop then runs to the longest path of the wave, each lane under its own `k < K`, and a lane without a
    //  contribution hands over DRT_ID_NONE and zeros)
    int Kw = K;
    if (Acc::kWave) {
#pragma unroll
        for (int off = DRT_WAVE / 2; off > 0; off >>= 1) {
            const int o2 = __shfl_xor(Kw, off);
            Kw = o2 > Kw ? o2 : Kw;
        }
    }
    for (int c0 = ((Kw - 1) / CH) * CH; c0 >= 0; c0 -= CH) {
        // prefix throughput at the start of this chunk (only for paths longer than a chunk)
        V3<R> T = mk<R>(R(1), R(1), R(1));
        // (a lane whose path ends before this chunk has nothing to rebuild -- and its LAST vertex may carry no colour id)
        for (int j = 0; j < c0 && c0 < K; ++j) {
            const TapeRec<R> tr = tape[(size_t)j * N + i];
            T = T * load_param<R, SMALL>(lds, params, (int)(tr.ids & 0xFFFFu)) * tr.m;
        }
        R Tx[CH], Ty[CH], Tz[CH], M[CH];
        uint32_t ID[CH];
        TapeRec<R> trs[CH];
        if (first_chunk && c0 == 0) {
            // vertices 0 .. CH - 1 were requested together with the path's vertex count (k_backward)
#pragma unroll
            for (int j = 0; j < CH; ++j)
                trs[j] = first_chunk[j];
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (c0 + j < K)
                    trs[j] = tape[(size_t)(c0 + j) * N + i];   // independent loads, all in flight
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
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
        for (int j = CH - 1; j >= 0; --j) {
            const int k = c0 + j;
            uint32_t ide = DRT_ID_NONE, idc = DRT_ID_NONE;
            V3<R> ve = mk<R>(R(0), R(0), R(0)), vc = ve;
            if (k < K) {
                const uint32_t cid = ID[j] & 0xFFFFu, eid = ID[j] >> 16;
                const R inv_pk = k >= a.min_bounces ? inv_p_rr : R(1);
                const V3<R> adj = g * mk<R>(Tx[j], Ty[j], Tz[j]);
                V3<R> Lk = mk<R>(R(0), R(0), R(0));
                if (eid != DRT_ID_NONE) {
                    ide = eid;
                    ve = adj * inv_pk;
                    Lk = load_param<R, SMALL>(lds, params, (int)eid) * inv_pk;
                }
                if (cid != DRT_ID_NONE) {
                    const V3<R> wgt = Ln * M[j];
                    idc = cid;
                    vc = adj * wgt;
                    Lk = Lk + load_param<R, SMALL>(lds, params, (int)cid) * wgt;
                }
                Ln = Lk;
            }
            if (Acc::kWave) {
                acc.add_wave(acc_lds, grad, ide, ve);
                acc.add_wave(acc_lds, grad, idc, vc);
            } else {
                if (ide != DRT_ID_NONE) acc.add(acc_lds, grad, ide, ve);
                if (idc != DRT_ID_NONE) acc.add(acc_lds, grad, idc, vc);
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

// ---- K6 (kernel; the tape walk and the accumulators it uses are defined above) ----
template <typename R, int NP>
#ifndef DRT_BACKWARD_GEN_MIN_BLOCKS
#define DRT_BACKWARD_GEN_MIN_BLOCKS 4   // blocks per CU K6's general accumulator (NP = 0: any number of parameters, an albedo per face) is compiled for: 128
                                        // registers + 88 B of scratch per lane (the compiler's own choice: 150, three waves per SIMD).  An albedo per face,
                                        // 1024 x 1024 x 8, ms per step, builds alternating: 0.79 -> 0.74 at four; five: 1.92 (tools/ab_kernel.py)
#endif
#ifndef DRT_BACKWARD_MIN_BLOCKS
#define DRT_BACKWARD_MIN_BLOCKS 4       // ... the register accumulators' kernel (NP = 4): config 4's K6, ms per step: 0.196 at four, 0.232 at three, 0.565 at five
#endif
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 && NP == 4) ? DRT_BACKWARD_MIN_BLOCKS : ((sizeof(R) == 4 && NP == 0) ? DRT_BACKWARD_GEN_MIN_BLOCKS : 1))
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
    for (uint32_t base = blockIdx.x * blockDim.x; base < a.n_paths; base += stride) {
        // (the waves of a block walk the loop whole: the general accumulator adds a wave at a time; a lane beyond the batch has
        //  no vertices)
        const bool valid = base + threadIdx.x < a.n_paths;
        const uint32_t i = valid ? base + threadIdx.x : a.n_paths - 1;
        // The first chunk of the tape is requested WITH the vertex count, not after it: one round trip
        // to memory per path instead of two (records beyond the path's end are read and ignored; the
        // rows exist for every depth below the cap).
        const int K = valid ? (int)nv[i] : 0;
        constexpr int CH = GradAcc<R, NP>::kWave ? DRT_TAPE_CHUNK_WAVE : DRT_TAPE_CHUNK;
        TapeRec<R> first[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (j < a.depth_cap)
                first[j] = tape[(size_t)j * N + i];
        V3<R> L0 = mk<R>(R(0), R(0), R(0));
        if (K > 0 || GradAcc<R, NP>::kWave)
            L0 = backward_path<R, SMALL>(a, lds, params, tape, N, i, K, path_seed<R>(a, adjoint, i, radiance_in), inv_p_rr, ga, acc, grad, first);
        if (lacc && valid) {
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
    static constexpr bool kWave = false;
    uint32_t param;
    V3<R> sum;
    __device__ inline void add(R (*)[DRT_BLOCK], double* __restrict__, uint32_t id, V3<R> v)
    {
        if (id == param)
            sum = sum + v;
    }
    __device__ inline void add_wave(R (*a)[DRT_BLOCK], double* __restrict__ g, uint32_t id, V3<R> v) { add(a, g, id, v); }
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

// ---- K7 ---------------------------------------------------------------------------------------
// grad[p] += sum over blocks of gpart[block][p] in a fixed order (deterministic); one block per row p
__global__ void __launch_bounds__(DRT_BLOCK)
k_gradreduce(const double* __restrict__ gpart, int n_blocks, int n_rows, double* __restrict__ grad, int row_stride,
             const unsigned short* __restrict__ slot_map = nullptr)
{
    // (slot_map, the one-launch kernels' general form: gpart's rows are table rows -- word p of the gradient vector, channel
    //  p % 3 of parameter p / 3, comes from row slot_map[p / 3] * 3 + p % 3, or from nowhere)
    __shared__ double red[DRT_BLOCK];
    const int p = blockIdx.x;
    int src = p;
    if (slot_map) {
        const int slot = slot_map[p / 3];
        src = slot == (int)DRT_SLOT_NONE ? n_rows : slot * 3 + p % 3;
    }
    if (src >= n_rows)
        return;
    double v = 0;
    for (int b = threadIdx.x; b < n_blocks; b += DRT_BLOCK)
        v += gpart[(size_t)b * row_stride + src];
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
