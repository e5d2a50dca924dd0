// drt_path.h -- k_path: K1 + K2 + K3 + K6 of a render in ONE launch (gfx950 / wave64).
//
// For scenes of analytic shapes whose paths end at a fixed depth (`-b D -p 1`, or a small max_depth) nothing
// has to leave the CU between the eye and the end of a path: a lane keeps ONE PIXEL, walks its samples one
// after the other and carries every path in registers --
//     ray (o, d), RNG key                                    Camera::sample, camera.hpp:51-60
//     T_k = prod_{j<k} colour_j m_j   (prefix throughput)    Pathtracer::scatter, pathtracer.hpp:91-115
//     L   = sum_k T_k E_k / p_k       (radiance of the path) Pathtracer::trace,   pathtracer.hpp:121-136
// and, when gradients are wanted (<= 4 parameters, the reference's scene has exactly 4, render.cpp:26-29), the
// TANGENTS of the throughput with respect to every parameter,
//     dT_k/dc_p,  updated per vertex:  dT' = dT * (colour m) + [colour is c_p] T m ,
// so that every emissive vertex adds  g * dT_p * E_k / p_k  to parameter p's gradient and  g * T_k / p_k  to its
// own emission parameter -- the very sums the reference's backward functors (vector.hpp:418-484) and
// VariableNode::backward's `m_grad += grad` (vector.hpp:185-188) produce by walking its graph in reverse; here
// they are accumulated in the order the path is traced, so no tape, no vertex count, no queue and no second
// kernel exist: per PATH the launch moves 0 bytes (per pixel and sample range: 24 bytes of radiance sums).
// The wavefront kernels of drt_kernels.h (queues in HBM, tape + K6) remain the general path: meshes, roulette-
// terminated paths of unbounded length, the unbiased operator, more than 4 parameters, gradient images.
//
// Work split: wave <-> (group of 64 consecutive pixels of the batch, range of `spr` samples); a block is four
// neighbouring groups of one range.  Pixel sums leave as f64 partials per range (summed in range order by
// k_film_parts: bitwise reproducible), gradients as per-block fp64 partials for K7 (fixed order), segment counts
// as one word per wave.
//
// Closest hit, f32: the scene's INTERSECTION PROGRAM (DevScene::prog*, built at upload, scene order kept so the
// first shape wins ties, pathtracer.hpp:80): one record per shape with a kind --
//     general plane   t = (o.n - off) * rcp(-(d.n))                        shape.hpp:49-59
//     axis plane      n = +-e_a exactly:  t = (s off - o_a) * rcp(d_a)      -- bit-identical to the general form
//                     (the products with 0 and +-1 are exact, rcp is odd), 2 instead of 9 VALU and the three
//                     rcp(d_a) are shared by all axis planes of the scene
//     sphere          half-b form of shape.hpp:78-103 (b = 2 b', disc = 4 disc': exact power-of-two scalings)
// The kinds are either read from the scene (the kind-sorted program in LDS: one counted loop per kind) or fixed at compile
// time (template SG, a KindSig: the shape loop fully unrolled, records in scalar registers, no branches at all) -- for
// the reference's own scene in the library, for any other analytic scene by hiprtc at run time (drt_jit.h).
#pragma once

#include "drt_kernels.h"

#define DRT_DRAW_TABLE (3 * DRT_MAX_DEPTH + 8)     // camera 2 (+1), then per vertex theta, phi and the next depth's roulette

struct PathArgs {
    // frame / sharding
    int32_t W, H, spp;
    int32_t shard, n_shards, band;
    // batch: pixels [p0, p0 + Pb) of the shard, samples [s0, s0 + Sb), cut into ranges of spr samples
    uint32_t Pb, p0, Sb, s0, spr, n_ranges, n_groups;
    // integrator
    int32_t min_bounces, depth_cap, cap_is_roulette, cap_draws;   // (cap_draws: BatchArgs)
    uint32_t rr_threshold, seed, rng_stream;   // rng_stream: drt_rng_stream(seed, 0)
    uint32_t regen_min;             // regenerating form: idle lanes it takes to run the camera code (see k_path)
    uint32_t shade_min, descend_min;   // k_path_mesh (drt_path_mesh.h): lanes it takes to run the vertex step / to keep the node loop going
    int32_t gimg_param;             // >= 0: the lanes' gradient sums of this parameter also leave per pixel (gradient image)
    uint32_t gen_rows, gen_clog2;   // NP = DRT_NP_ANY: rows of the gradient tables (3 per parameter that requires a gradient), log2 of the copies a wave keeps of each
    uint32_t hist_lds, hist_stride; // ... full history words a thread keeps in LDS (the rest: global memory), threads of the grid
    uint32_t gimg_row, pad_gen;     // ... the table row of the gradient image's parameter (NC == 1), DRT_SLOT_NONE: it requires no gradient
    unsigned long long hist_ovf;    // ... the history's global part (uint32_t[words][hist_stride]), 0: none
    double p_rr, inv_p_rr;          // 1 - absorb and its reciprocal (pathtracer.hpp:130)
    // camera
    double eye[3], fwd[3], right[3], up[3];
    double tan_half, aspect, inv_W, inv_H;
    // what the f32 kernels use of the doubles above, cast on the host: a scalar register each instead of a v_cvt_f32_f64 result in a vector one
    // (at the END of the record: the f64 kernels' argument offsets stay what their register allocation was tuned at)
    float p_rr_f, inv_p_rr_f;
    float eye_f[3], fwd_f[3], right_f[3], up_f[3], cs_step_f, ct_step_f;   // (cs = cs0 + u1 cs_step: camera.hpp:53-58 in CameraLane's form)
};

__device__ inline uint32_t path_global_pixel(const PathArgs& a, uint32_t lp)
{
    uint32_t ly = lp / (uint32_t)a.W, x = lp - ly * (uint32_t)a.W;
    uint32_t y = ly;
    if (a.n_shards > 1) {
        uint32_t b = ly / (uint32_t)a.band, r = ly - b * (uint32_t)a.band;
        y = (b * (uint32_t)a.n_shards + (uint32_t)a.shard) * (uint32_t)a.band + r;
    }
    return y * (uint32_t)a.W + x;
}

template <typename SG>
__device__ inline HitRec<float> path_closest_hit(const DevScene<float>* __restrict__ sc, const ProgRecs<SG::n>& recs, float4 ra, float2 rb)
{
    return closest_hit_prog<SG>(sc, recs, mk<float>(ra.x, ra.y, ra.z), mk<float>(ra.w, rb.x, rb.y));
}
template <typename SG>
__device__ inline HitRec<double> path_closest_hit(const DevScene<double>* __restrict__ sc, const ProgRecs<SG::n, double>& recs, double4 ra, double2 rb)
{
    if (SG::n > 0)                                                 // a compiled-in program in f64
        return closest_hit_sig<SG, double>(recs, mk<double>(ra.x, ra.y, ra.z), mk<double>(ra.w, rb.x, rb.y));
    const double4 ra1[1] = {ra};
    const double2 rb1[1] = {rb};
    HitRec<double> h1[1];
    closest_hit_n<double, 1>(sc, sc->n_shapes, ra1, rb1, h1);      // any other scene: the literal loop
    return h1[0];
}

// the kinds of the reference's scene (render.cpp:39-47): sphere, sphere, -x, general (1, 0, 0.1), -z, +z, +y, -y, sphere
#define DRT_SIG_CORNELL                                                                                                   \
    ((unsigned long long)DRT_PK_SPHERE | (unsigned long long)DRT_PK_SPHERE << 3 | (unsigned long long)DRT_PK_AX << 6 |    \
     (unsigned long long)DRT_PK_PLANE << 9 | (unsigned long long)DRT_PK_AZ << 12 | (unsigned long long)DRT_PK_AZ << 15 |  \
     (unsigned long long)DRT_PK_AY << 18 | (unsigned long long)DRT_PK_AY << 21 | (unsigned long long)DRT_PK_SPHERE << 24)
#define DRT_NSIG_CORNELL 9
typedef KindSig<DRT_SIG_CORNELL, 0ull, 0ull, 0ull, DRT_NSIG_CORNELL> SigCornell;

// (DRT_PATH_LDS_PARAMS, drt_device.h: parameters staged in LDS; more: read from L2)

// the part of the scene a vertex of the one-launch kernels needs, compact (SceneLds carries the whole DevScene and 256
// parameters: 10.4 KB; this is 5.2 KB in f32 -- a block per CU more for the regenerating k_path)
template <typename R>
struct PathSceneLds {
    struct {
        int n_shapes, n_materials, n_emitters, n_params;
        DevShape<R> shapes[DRT_MAX_SHAPES];
        DevMaterial<R> materials[DRT_MAX_MATERIALS];
        int emitter_param[DRT_MAX_EMITTERS];
        int flat[DRT_MAX_SHAPES];
    } sc;
    R params[DRT_PATH_LDS_PARAMS * 3];
#ifdef DRT_USER_SHAPES
    R user_q[DRT_MAX_SHAPES][4];           // caller-defined shapes: values 4..7 of their records (the normal needs the whole record)
#endif
};

template <typename R, bool ALL_LDS = false>
__device__ inline V3<R> load_param(const PathSceneLds<R>& lds, const R* __restrict__ params, int id)
{
    if (ALL_LDS || id < DRT_PATH_LDS_PARAMS)
        return mk<R>(lds.params[id * 3], lds.params[id * 3 + 1], lds.params[id * 3 + 2]);
    return mk<R>(params[id * 3], params[id * 3 + 1], params[id * 3 + 2]);
}

template <typename R>
__device__ inline void stage_path_scene(PathSceneLds<R>& lds, const DevScene<R>* __restrict__ sc, const R* __restrict__ params)
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
    for (int i = threadIdx.x; i < ns; i += blockDim.x)
        lds.sc.flat[i] = sc->flat[i];
#ifdef DRT_USER_SHAPES
    for (int i = threadIdx.x; i < ns * 4; i += blockDim.x)
        lds.user_q[i >> 2][i & 3] = sc->user_q[i >> 2][i & 3];
#endif
    const int np = sc->n_params < DRT_PATH_LDS_PARAMS ? sc->n_params : DRT_PATH_LDS_PARAMS;
    for (int i = threadIdx.x; i < np * 3; i += blockDim.x)
        lds.params[i] = params[i];
    __syncthreads();
}


// per-lane gradient state: NP parameters (0 = none), of which only the first NC can be a BxDF's colour (the others are
// emission-only parameters: the reference's scene has three albedos and one emission, render.cpp:26-29).
//
// The throughput is a PRODUCT, T = prod_j colour_{p_j} m_j, so its tangent with respect to a colour parameter needs no
// per-vertex update at all (rounds 2-3 carried dT/dc_p per parameter and moved it with 9 fma + 9 selects per bounce):
//     dT_ch / dc_{p,ch} = T_ch n_p / c_{p,ch} ,      n_p = how many vertices of the path scattered on colour p
// -- one 8-bit counter per parameter (a path has at most DRT_MAX_DEPTH = 64 vertices), one packed add per bounce.  A colour
// channel that is ZERO (the reference's red = (0.5, 0, 0), render.cpp:26) has no quotient: the lane's T therefore leaves the
// zero factors out (they are replaced by 1: `colnz`) and counts them per channel instead (`zc`, 8 bits each):
//     T_real,ch       = zc_ch == 0 ? T_ch : 0
//     dT_ch/dc_{p,ch} = c_{p,ch} != 0 ? T_real,ch n_p / c_{p,ch}
//                                     : (n_p == 1 and zc_ch == 1 ? T_ch : 0)     (d/dc of c^n at 0: 1 for n = 1, else 0)
// Evaluated where a path meets a light: once per sample.  Same sums as the reference's backward functors
// (vector.hpp:418-484) in closed form; the image is unchanged bit for bit (a channel without zero factors sees the very
// same multiplications), gradients to f32 rounding.  |c| < 1e-18 counts as zero (its quotient would overflow).
template <typename R>
struct TangentLds {              // per colour parameter, wave-uniform except where indexed by the lane's colour id
    R colnz[DRT_FAST_PARAMS][4];         // the colour with zero channels replaced by 1
    R invc[DRT_FAST_PARAMS][4];          // 1 / c per channel, 0 where the channel is zero
    uint32_t inc[DRT_FAST_PARAMS][4];    // counter increments of a bounce on this colour: n_p (low, high word), zc; [3] = zero-channel bits
};

template <typename R, typename SL>
__device__ inline void stage_tangents(TangentLds<R>& tl, const SL& lds)
{
    // (after stage_scene's barrier; parameters beyond the scene's own read as (1, 1, 1))
    if (threadIdx.x < DRT_FAST_PARAMS) {
        const int p = threadIdx.x;
        const bool in = p < lds.sc.n_params;
        uint32_t zinc = 0, zbits = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const R c = in ? lds.params[p * 3 + ch] : R(1);
            const bool zero = abs_r(c) < R(1e-18);
            tl.colnz[p][ch] = zero ? R(1) : c;
            tl.invc[p][ch] = zero ? R(0) : R(1) / c;
            zinc |= zero ? 1u << (8 * ch) : 0u;
            zbits |= zero ? 1u << ch : 0u;
        }
        tl.colnz[p][3] = R(0);
        tl.invc[p][3] = R(0);
        tl.inc[p][0] = p < 4 ? 1u << (8 * p) : 0u;
        tl.inc[p][1] = p >= 4 ? 1u << (8 * (p - 4)) : 0u;
        tl.inc[p][2] = zinc;
        tl.inc[p][3] = zbits;
    }
    __syncthreads();
}

template <typename R, int NP, int NC>
struct Tangents {
    uint32_t cnt[NC > 4 ? 2 : 1];   // n_p of the current path, 8 bits per colour parameter
    uint32_t zc;                    // zero factors met per channel, 8 bits each
    // The lane's gradient sums, NP x 3 values, live in an LDS column of the block (acc[row][thread]: conflict-free), not
    // in registers: they are touched once per SAMPLE (where the path meets a light), and twelve registers held across the
    // bounce loop cost the kernel its sixth wave per SIMD (96 -> 80 VGPRs; as scratch spills, which is what the compiler
    // makes of them when told to fit six waves, 0.728 -> 0.704 ms on config 3; as an LDS column: see HISTORY.md 3a).
    R* acc;
    __device__ inline V3<R> acc_get(int p) const { return mk<R>(acc[(p * 3) * DRT_BLOCK], acc[(p * 3 + 1) * DRT_BLOCK], acc[(p * 3 + 2) * DRT_BLOCK]); }
    __device__ inline void acc_set(int p, V3<R> v) { acc[(p * 3) * DRT_BLOCK] = v.x; acc[(p * 3 + 1) * DRT_BLOCK] = v.y; acc[(p * 3 + 2) * DRT_BLOCK] = v.z; }
    __device__ inline void new_path() { cnt[0] = 0; if (NC > 4) cnt[NC > 4 ? 1 : 0] = 0; zc = 0; }
};

// ---- ANY number of parameters (NP = DRT_NP_ANY) -------------------------------------------------------------------------
// The reference differentiates with respect to every Vector<T,3,true> of the scene, however many (vector.hpp:185-191;
// render.cpp:26-29 has four because its scene is small).  Counters and an LDS column per parameter (above) stop at eight.
// The general form keeps the same closed form -- dT/dc_p = T n_p / c_p, i.e. every vertex j that scattered on a colour adds
// g T E / (p_k c_{id_j}) to the row of ITS colour once the path meets a light -- and remembers WHICH colour every vertex
// scattered on instead of counting per colour:
//   history   8 bits per vertex (parameter ids < DRT_PATH_LDS_PARAMS = 136; 0xFF = no vertex): the last <= 4 in a register,
//             full words in an LDS column of the thread (dynamic shared memory: as many words as fit without costing the
//             kernel a block per CU -- four in the lockstep kernel: paths of 16 vertices; none in the regenerating ones),
//             deeper ones in a column of global memory (written once per four vertices, read where the path meets a light);
//   tables    per WAVE, in LDS: a row per channel of every parameter that requires a gradient (DevScene::grad_slot), in
//             2^clog2 copies (lane & (copies - 1): same-address atomics serialise), added to with ds_add -- one wave's adds
//             reach its own table in program order, so the sums do not depend on how the waves of the chip were scheduled;
//   per id    GenLds: the colour with zero channels replaced by 1 + its zero-count increments, 1 / c + row and zero-channel bits
//             (one ds_read_b128 each).
// A zero channel (red = (0.5, 0, 0), render.cpp:26): vertex j's own factor is the only zero of the channel iff zc_ch == 1, and
// then d/dc of that channel is T_ch (the product WITHOUT the zero factor, which is what the lane's T holds) -- else 0.
#define DRT_NP_ANY (-1)
// Blocks per CU (= waves per SIMD) the f32 lockstep k_path is compiled for, form by form (ms per launch on config 3's frame, the builds
// alternating in one process: profiles/r06_waves_per_simd.txt; five against six: r06_scratch_vs_waves.txt)
#ifndef DRT_LOCKSTEP_MIN_BLOCKS
#define DRT_LOCKSTEP_MIN_BLOCKS 7       // diffuse, <= 4 parameters: 72 registers.  Five (96 registers) 0.716, six (80) 0.695-0.702, seven 0.673, eight (64 + scratch) 0.694-0.702
#endif
#ifndef DRT_LOCKSTEP_GEN_MIN_BLOCKS
#define DRT_LOCKSTEP_GEN_MIN_BLOCKS 6   // its general form (any number of parameters): 77-80 registers; seven 0.725 -> 0.736
#endif
#ifndef DRT_LOCKSTEP_SPEC_MIN_BLOCKS
#define DRT_LOCKSTEP_SPEC_MIN_BLOCKS 6  // with the glossy lobe (config 5), <= 4 parameters: 78-80 registers; config 5's shape (2048 x 2048 x 16, depth 16): five 7.92, six 7.79,
                                        // seven 7.81 (tools/ab_kernel.py)
#endif
#ifndef DRT_REGEN_MIN_BLOCKS
#define DRT_REGEN_MIN_BLOCKS 5       // blocks per CU the f32 regenerating diffuse k_path is compiled for
#endif
#ifndef DRT_F64_MIN_BLOCKS
#define DRT_F64_MIN_BLOCKS 4         // blocks per CU the f64 lockstep diffuse k_path is compiled for: 128 registers instead of 142-145, four waves per
                                     // SIMD instead of three (config 3 in f64: 1.938 -> 1.890 ms, an albedo per shape 2.19 -> 1.97; 3 and 5: no gain)
#endif
#ifndef DRT_GEN_TABLE
#define DRT_GEN_TABLE 408            // elements of a wave's gradient table (rows x copies; 408 = 3 x DRT_PATH_LDS_PARAMS: one copy of every row at least)
#endif
template <bool B, typename X, typename Y> struct PickT { typedef X T; };
template <typename X, typename Y> struct PickT<false, X, Y> { typedef Y T; };
struct NoLds { int unused; };
// The tables are fp64 in the f32 kernels too (ds_add_f64): a table sums what 64 lanes x their samples add, and ONE heavy sample
// (the unbiased operator's L' g / pdf; a roulette-boosted path) in an f32 sum costs every later add its low bits -- measured
// with f32 tables: 3.4e-3 of the largest component on the 12-parameter unbiased fixture, where per-lane f32 sums give 3e-6.
template <typename R> struct GenAcc { typedef double T; };
#ifdef DRT_GEN_F32
template <> struct GenAcc<float> { typedef float T; };
#endif

template <typename R>
struct GenLds {
    R colnz[DRT_PATH_LDS_PARAMS][4];   // colour, zero channels replaced by 1 | [3]: zero-count increments, 8 bits per channel (pid_pack)
    R invc[DRT_PATH_LDS_PARAMS][4];    // 1 / c per channel, 0 where the channel is zero | [3]: table row | zero-channel bits << 16 (pid_pack)
};

template <typename R, typename SL>
__device__ inline void stage_gen(GenLds<R>& gl, const SL& lds, const DevScene<R>* __restrict__ sc)
{
    // (after stage_path_scene's barrier; ids beyond the scene's own read as (1, 1, 1) without a row)
    for (int p = threadIdx.x; p < DRT_PATH_LDS_PARAMS; p += blockDim.x) {
        const bool in = p < lds.sc.n_params;
        uint32_t zinc = 0, zbits = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const R c = in ? lds.params[p * 3 + ch] : R(1);
            const bool zero = abs_r(c) < R(1e-18);
            gl.colnz[p][ch] = zero ? R(1) : c;
            gl.invc[p][ch] = zero ? R(0) : R(1) / c;
            zinc |= zero ? 1u << (8 * ch) : 0u;
            zbits |= zero ? 1u << ch : 0u;
        }
        const uint32_t slot = in ? (uint32_t)sc->grad_slot[p] : DRT_SLOT_NONE;
        gl.colnz[p][3] = pid_pack(R(0), zinc);
        gl.invc[p][3] = pid_pack(R(0), slot | zbits << 16);
    }
    __syncthreads();
}

template <typename R, int NC>
struct Tangents<R, DRT_NP_ANY, NC> {
    typedef typename GenAcc<R>::T GT;
    uint32_t cur;                   // the colours of the last <= 4 vertices, shifted in from the top (0xFF: none)
    uint32_t zc;                    // zero factors met per channel, 8 bits each
    uint32_t nv;                    // history entries of the current path (wave-uniform in the lockstep kernel)
    uint32_t* hist;                 // the thread's column of full history words: word w < hist_lds at hist[w * DRT_BLOCK] (LDS),
    uint32_t* hist_ovf;             // ... the others at hist_ovf[(w - hist_lds) * hist_stride] (global)
    uint32_t hist_lds, hist_stride;
    const GenLds<R>* gl;
    GT* table;                      // the wave's table, at the lane's copy: element (row, copy) at table[(row << clog2)]
    uint32_t clog2;
    R* acc;                         // (unused: the register / column forms' sums)
    // NC == 1, the gradient IMAGE (README.md:142-145; lockstep form, a lane IS a pixel): what the lane itself adds to the row of
    // the image's parameter, next to the wave's table
    V3<R> gsum;
    uint32_t gimg_row;
    __device__ inline void new_path() { cur = 0xFFFFFFFFu; zc = 0; nv = 0; }
    __device__ inline void store_word(uint32_t w, uint32_t v)
    {
        if (w < hist_lds) hist[w * DRT_BLOCK] = v;
        else hist_ovf[(size_t)(w - hist_lds) * hist_stride] = v;
    }
    __device__ inline uint32_t load_word(uint32_t w) const
    {
        return w < hist_lds ? hist[w * DRT_BLOCK] : hist_ovf[(size_t)(w - hist_lds) * hist_stride];
    }
    // one more vertex: `id` = the colour it scattered on, 0xFF = none (the path ended there).  UNIFORM: every lane of the wave
    // pushes at every bounce (the lockstep kernel: nv stays a scalar); else only lanes with `live`.
    template <bool UNIFORM>
    __device__ inline void push(bool live, uint32_t id)
    {
        const uint32_t shifted = (cur >> 8) | (id << 24);
        if (UNIFORM) {
            cur = shifted;
            ++nv;
            if ((nv & 3u) == 0u) {
                store_word((nv >> 2) - 1u, cur);
                cur = 0xFFFFFFFFu;
            }
        } else {
            cur = live ? shifted : cur;
            nv += live ? 1u : 0u;
            if (live && (nv & 3u) == 0u) {
                store_word((nv >> 2) - 1u, cur);
                cur = 0xFFFFFFFFu;
            }
        }
    }
    __device__ inline void add(uint32_t row, V3<R> v)
    {
        if constexpr (NC == 1) {
            const bool mine = row == gimg_row;
            gsum = mk<R>(gsum.x + (mine ? v.x : R(0)), gsum.y + (mine ? v.y : R(0)), gsum.z + (mine ? v.z : R(0)));
        }
        GT* t = table + ((row * 3u) << clog2);
        atomicAdd(t, (GT)v.x);
        atomicAdd(t + (1u << clog2), (GT)v.y);
        atomicAdd(t + (2u << clog2), (GT)v.z);
    }
    // the four vertices of one history word: each adds  TgE / c  (U where its own channel is the zero one) to its colour's row
    __device__ inline void add_word(uint32_t wd, V3<R> TgE, V3<R> U)
    {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t id = (wd >> (8 * b)) & 0xFFu;
            const bool valid = id != 0xFFu;
            if (wave_any(valid)) {
                if (valid) {
                    const R* rec = gl->invc[id];
                    const R ix = rec[0], iy = rec[1], iz = rec[2];
                    const uint32_t bits = pid_unpack(rec[3]);
                    const uint32_t row = bits & 0xFFFFu;
                    if (row != DRT_SLOT_NONE)
                        add(row, mk<R>((bits & 0x10000u) ? U.x : TgE.x * ix, (bits & 0x20000u) ? U.y : TgE.y * iy,
                                       (bits & 0x40000u) ? U.z : TgE.z * iz));
                }
            }
        }
    }
};

// what a block of a general-form kernel keeps in LDS, and the two ends of its life
template <typename R>
struct GenBlock {
    GenLds<R> gl;
    typename GenAcc<R>::T table[DRT_BLOCK / DRT_WAVE][DRT_GEN_TABLE];
};
// (before stage_path_scene's barrier: the tables start at zero)
template <typename R>
__device__ inline void gen_zero(GenBlock<R>& gb)
{
    typedef typename GenAcc<R>::T GT;
    for (uint32_t i = threadIdx.x; i < (DRT_BLOCK / DRT_WAVE) * DRT_GEN_TABLE; i += DRT_BLOCK)
        (&gb.table[0][0])[i] = GT(0);
}
// (after it: the per-id records; the thread's view of its wave's table)
template <typename R, int NC, typename SL>
__device__ inline void gen_begin(GenBlock<R>& gb, const SL& lds, const DevScene<R>* __restrict__ sc, const PathArgs& a, uint32_t* hist,
                                 uint32_t* hist_ovf, Tangents<R, DRT_NP_ANY, NC>& tg)
{
    stage_gen(gb.gl, lds, sc);
    tg.gl = &gb.gl;
    tg.hist = hist + threadIdx.x;
    tg.hist_ovf = hist_ovf + (size_t)blockIdx.x * DRT_BLOCK + threadIdx.x;
    tg.gsum = mk<R>(R(0), R(0), R(0));
    tg.gimg_row = a.gimg_row;
    tg.hist_lds = a.hist_lds;
    tg.hist_stride = a.hist_stride;
    tg.clog2 = a.gen_clog2;
    tg.table = &gb.table[threadIdx.x / DRT_WAVE][threadIdx.x & ((1u << a.gen_clog2) - 1u)];
    tg.acc = nullptr;
    tg.new_path();
}
// the waves' tables -> the block's row sums in fp64: copies, then waves, in a fixed order; K7 / the finishing launch adds the blocks
template <typename R>
__device__ inline void gen_finish(const GenBlock<R>& gb, const PathArgs& a, double* __restrict__ gpart)
{
    __syncthreads();
    const uint32_t copies = 1u << a.gen_clog2;
    for (uint32_t r = threadIdx.x; r < a.gen_rows; r += DRT_BLOCK) {
        double v = 0;
        for (int ww = 0; ww < DRT_BLOCK / DRT_WAVE; ++ww)
            for (uint32_t c = 0; c < copies; ++c)
                v += (double)gb.table[ww][(r << a.gen_clog2) + c];
        gpart[(size_t)blockIdx.x * a.gen_rows + r] = v;
    }
}

// an emissive vertex reached with prefix throughput T: radiance and gradients
//   L     += T E / p_k                                   (pathtracer.hpp:113-114, 133)
//   d/dc_p += g dT_p E / p_k      d/dE += g T / p_k       (vector.hpp:418-484 in closed form)
// LOSS (DRT_RENDER_LOSS_L2, the end of a path only): `g` holds the lane's TARGET pixel and the seed is the derivative of the
// sample's own squared error, 2 (L - target), with L the path's radiance INCLUDING this emission -- final where the path ends
// on a light without BxDF, which is the only emissive vertex of a path in the scenes this form is used for.
template <typename R, int NP, int NC, bool LOSS = false, typename SL = PathSceneLds<R>>
__device__ inline void add_emission(const SL& lds, const TangentLds<R>& tl, const R* __restrict__ params, uint32_t eid, R inv_pk,
                                    V3<R> T, V3<R> g, V3<R>& L, Tangents<R, NP, NC>& tg)
{
    const V3<R> E = load_param<R, (NP != 0)>(lds, params, (int)eid) * inv_pk;
    V3<R> Tr = T;
    if (NC > 0 || NP == DRT_NP_ANY)  // a channel that met a zero colour is dark
        Tr = mk<R>((tg.zc & 0xFFu) ? R(0) : T.x, (tg.zc & 0xFF00u) ? R(0) : T.y, (tg.zc & 0xFF0000u) ? R(0) : T.z);
    L = L + Tr * E;
    if constexpr (NP == DRT_NP_ANY) {
        // any number of parameters: the light's own row, then every vertex of the path's history adds to its colour's row
        if (LOSS)
            g = mk<R>(R(2) * (L.x - g.x), R(2) * (L.y - g.y), R(2) * (L.z - g.z));
        const V3<R> gE = g * E;
        const uint32_t ebits = pid_unpack(tg.gl->invc[eid < DRT_PATH_LDS_PARAMS ? eid : 0][3]);
        if ((ebits & 0xFFFFu) != DRT_SLOT_NONE && eid < DRT_PATH_LDS_PARAMS)
            tg.add(ebits & 0xFFFFu, g * Tr * inv_pk);
        const V3<R> TgE = Tr * gE;
        const V3<R> U = mk<R>((tg.zc & 0xFFu) == 0x1u ? T.x * gE.x : R(0), (tg.zc & 0xFF00u) == 0x100u ? T.y * gE.y : R(0),
                              (tg.zc & 0xFF0000u) == 0x10000u ? T.z * gE.z : R(0));
        const uint32_t nw = tg.nv >> 2;
        for (uint32_t w = 0; wave_any(w < nw); ++w)
            tg.add_word(w < nw ? tg.load_word(w) : 0xFFFFFFFFu, TgE, U);
        tg.add_word(tg.cur, TgE, U);
    } else
    if constexpr (NP > 0) {
        if (NC > 0)
            asm volatile("" ::: "memory");    // (keeps the reads of `tl` below where they are: see there)
        if (LOSS)
            g = mk<R>(R(2) * (L.x - g.x), R(2) * (L.y - g.y), R(2) * (L.z - g.z));
        const V3<R> gE = g * E, gT = g * Tr * inv_pk;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bool own = eid == (uint32_t)p;
            const V3<R> ap = tg.acc_get(p);
            const V3<R> a0 = mk<R>(ap.x + (own ? gT.x : R(0)), ap.y + (own ? gT.y : R(0)), ap.z + (own ? gT.z : R(0)));
            if (p < NC) {
                const uint32_t n = (tg.cnt[p >> 2] >> (8 * (p & 3))) & 0xFFu;
                const R nf = (R)(int)n;
                // (read where they are used, once per sample: hoisted out of the sample loop these twelve wave-uniform words
                //  cost the kernel its fifth wave per SIMD)
                V3<R> dT = mk<R>(Tr.x * (nf * tl.invc[p][0]), Tr.y * (nf * tl.invc[p][1]), Tr.z * (nf * tl.invc[p][2]));
                const uint32_t zb = __builtin_amdgcn_readfirstlane(tl.inc[p][3]);
                if (zb) {
                    const bool one = n == 1u;
                    if (zb & 1u) dT.x = (one && (tg.zc & 0xFFu) == 0x1u) ? T.x : R(0);
                    if (zb & 2u) dT.y = (one && (tg.zc & 0xFF00u) == 0x100u) ? T.y : R(0);
                    if (zb & 4u) dT.z = (one && (tg.zc & 0xFF0000u) == 0x10000u) ? T.z : R(0);
                }
                tg.acc_set(p, mk<R>(fma_r(dT.x, gE.x, a0.x), fma_r(dT.y, gE.y, a0.y), fma_r(dT.z, gE.z, a0.z)));
            } else
                tg.acc_set(p, a0);
        }
    }
}

template <typename R>
struct CameraLane {            // per-lane camera constants (the lane's pixel does not change over its samples)
    R cs0, ct0;                // (2 x / W - 1) aspect tan(vfov / 2) and (2 y / H - 1) tan(vfov / 2) at the pixel's corner
};

// ---- one bounce of one path (every lane executes it; `live` says whether the lane's path is still going) -----------
// pk, inv_pk, n_theta, next_rr, next_cap describe the depth of the vertex: wave-uniform scalars in the fixed-depth
// kernel (all lanes at the same depth), per-lane values in the regenerating one.  On return: ra/rb = the next ray,
// T/dT moved on iff `alive`; `on_light` = the path ended on a light without BxDF (pathtracer.hpp:38-39: f = 0) whose
// emission parameter is `light` -- the caller adds that emission (the fixed-depth kernel once per sample, after its
// bounce loop, for all lanes together; the regenerating kernel when the lane's path ends).
// what the unbiased backward needs to know about a vertex (it re-samples a fresh direction there)
template <typename R>
struct PathVertex {
    V3<R> P, nrm, d;           // point, normal, direction of the ray that arrived
    uint32_t ids;              // colour | emission << 16 parameter ids
    int material;              // index of the BxDF's material record (valid when scattered)
    bool hit, scattered;       // the ray hit something / something with a BxDF
};

template <typename R, bool SPEC, int NP, int NC, typename SG>
__device__ inline void path_bounce(const PathArgs& a, const PathSceneLds<R>& lds, const TangentLds<R>& tl, const DevScene<R>* __restrict__ sc,
                                   const R* __restrict__ params, const ProgRecs<SG::n, R>& recs, uint32_t key,
                                   R pk, R inv_pk, uint32_t n_theta, bool next_rr, bool next_cap, bool live, V3<R> g,
                                   typename Q4<R>::T& ra, typename Q2<R>::T& rb, V3<R>& T, V3<R>& L, Tangents<R, NP, NC>& tg,
                                   bool& alive, bool& capped, bool& on_light, uint32_t& light, PathVertex<R>* vo = nullptr,
                                   bool last = false, const float* __restrict__ seed_px = nullptr, const uint32_t* ih = nullptr)
{
    // (ih, the forms whose lanes stand at their own depths: the draw indices' hash rounds h(n) as a table in LDS -- where all
    //  lanes of a wave share the index, the lockstep kernel, the scalar unit computes h(n) and the table would only cost)
    auto draw = [&](uint32_t n) { return ih ? drt_rng_combine(ih[n], key) : rng_draw(a.rng_stream, key, n); };
    // (seed_px, the regenerating form: where the pixel's adjoint seed stands in the caller's image -- read only where a vertex
    //  emits, instead of carrying it in registers for a lane that changes pixel with every path)
    // `last` (wave-uniform; the lockstep kernel at the deepest vertex a path can have): no lane's path goes on from here, so
    // nothing is sampled -- the reference does sample a direction there, and the trace() it hands it to is absorbed before it
    // casts a ray (pathtracer.hpp:128): a factor of exactly 0.  What is left of the bounce is the hit, the light it may have
    // ended on, and the count of paths a user cap (not the roulette) cut short.  Same results bit for bit, ~110 of the
    // bounce's ~280 vector instructions less: 5 % of a depth-8 frame.
    const HitRec<R> h = path_closest_hit<SG>(sc, recs, ra, rb);
    const bool hit = live && h.prim >= 0;
    const int prim = h.prim >= 0 ? h.prim : 0;                        // (a miss reads record 0, uses nothing of it)
    const V3<R> o = mk<R>(ra.x, ra.y, ra.z);
    const V3<R> d = mk<R>(ra.w, rb.x, rb.y);
    const V3<R> P = o + d * h.t;                                      // pathtracer.hpp:83
    const DevShape<R>& sh = lds.sc.shapes[prim];
    const uint32_t ids = (uint32_t)sh.pad;                            // colour | emission << 16 parameter ids
    const uint32_t cid = ids & 0xFFFFu, eid = ids >> 16;
    const bool has_bxdf = cid != DRT_ID_NONE, emits = hit && eid != DRT_ID_NONE;
    // emission, pathtracer.hpp:113-114: a shape with BxDF AND emitter (rare) adds it here, a pure light is the caller's
    on_light = emits && !has_bxdf;
    light = eid;
    if (last && !vo) {
        if (wave_any(emits && has_bxdf)) {
            if (emits && has_bxdf) {
                if (NP != 0 && seed_px)
                    g = mk<R>((R)seed_px[0], (R)seed_px[1], (R)seed_px[2]);
                add_emission<R, NP, NC>(lds, tl, params, eid, inv_pk, T, g, L, tg);
            }
        }
        alive = false;
        capped = false;
        if (!a.cap_is_roulette) {                                      // (a user max_depth: had the roulette let the path live?)
            const bool rr_kills = next_rr && draw(n_theta + 2) < a.rr_threshold;
            capped = hit && has_bxdf && !rr_kills;
        }
        return;
    }
    const V3<R> ctr = mk<R>(sh.p[0], sh.p[1], sh.p[2]);
    // (f32: (P - c) / r with 1 / r from a table would save the rsq and the select -- and moves sphere normals by a few ulp,
    //  which flips one grazing path of the 64 x 48 x 8 smoke frame: measured, not kept; the literal form stays)
    const V3<R> nsph = normalize(P - ctr);                            // shape.hpp:105-106
    const bool is_plane = sh.type == DRT_SHAPE_PLANE;                 // shape.hpp:58-59: the normal as stored
    V3<R> nrm = mk<R>(is_plane ? ctr.x : nsph.x, is_plane ? ctr.y : nsph.y, is_plane ? ctr.z : nsph.z);
#ifdef DRT_USER_SHAPES
    if (sh.type >= DRT_SHAPE_USER) {                                  // a caller-defined kind: its own normal(point) (shape.hpp:22)
        const R p8[8] = {sh.p[0], sh.p[1], sh.p[2], sh.p[3], lds.user_q[prim][0], lds.user_q[prim][1], lds.user_q[prim][2], lds.user_q[prim][3]};
        nrm = user_normal<R>(sh.type - DRT_SHAPE_USER, p8, P);
    }
#endif
    if (vo) {
        vo->P = P; vo->nrm = nrm; vo->d = d;
        vo->ids = ids;
        vo->material = sh.material;
        vo->hit = hit;
        vo->scattered = hit && has_bxdf;
    }
    if (wave_any(emits && has_bxdf)) {
        if (emits && has_bxdf) {
            if (NP != 0 && seed_px)
                g = mk<R>((R)seed_px[0], (R)seed_px[1], (R)seed_px[2]);
            add_emission<R, NP, NC>(lds, tl, params, eid, inv_pk, T, g, L, tg);
        }
    }
    // the BxDF: sample, evaluate (pathtracer.hpp:91-111)
    const DevMaterial<R>& m = lds.sc.materials[has_bxdf ? sh.material : 0];
    V3<R> wo;
    R q, bs;
    sample_bxdf<R, SPEC>(m, nrm, d, draw(n_theta), draw(n_theta + 1), wo, q, bs);
    const R c = dot(nrm, wo);                                         // pathtracer.hpp:103
    const R mk_ = div_r(bs * c, q * pk);                              // T_{k+1} = T_k * colour * m_k (f32: v_rcp, 1 ulp)
    // roulette / cap of the next depth (pathtracer.hpp:128)
    const bool rr_kills = next_rr && draw(n_theta + 2) < a.rr_threshold;
    alive = hit && has_bxdf && !next_cap && !rr_kills;
    capped = hit && has_bxdf && next_cap && !a.cap_is_roulette && !rr_kills;
    // the throughput moves on only in lanes whose path goes on (the others stay frozen for the light's turn); with
    // gradients it leaves zero colour channels out and counts them, and counts the bounce for its colour (see Tangents)
    const int cidx = has_bxdf ? (int)cid : 0;
    V3<R> col;
    if constexpr (NP == DRT_NP_ANY) {
        const R* rec = tg.gl->colnz[cidx < DRT_PATH_LDS_PARAMS ? cidx : 0];
        col = mk<R>(rec[0], rec[1], rec[2]);
        tg.zc += alive ? pid_unpack(rec[3]) : 0u;
        if (ih)
            tg.template push<false>(alive, cid);      // (a lane on its own: only vertices the path goes on from)
        else
            tg.template push<true>(live, alive ? cid : 0xFFu);
    } else
        col = NC > 0 ? mk<R>(tl.colnz[cidx][0], tl.colnz[cidx][1], tl.colnz[cidx][2]) : load_param<R, (NP > 0)>(lds, params, cidx);
    const V3<R> cmv = col * mk_;
    const V3<R> cm = mk<R>(alive ? cmv.x : R(1), alive ? cmv.y : R(1), alive ? cmv.z : R(1));
    if constexpr (NC > 0 && NP != DRT_NP_ANY) {
        tg.cnt[0] += alive ? tl.inc[cidx][0] : 0u;
        if (NC > 4)
            tg.cnt[NC > 4 ? 1 : 0] += alive ? tl.inc[cidx][1] : 0u;
        tg.zc += alive ? tl.inc[cidx][2] : 0u;
    }
    T = T * cm;
    const V3<R> no = P + wo * R(1e-3);                                // pathtracer.hpp:99
    ra.x = no.x; ra.y = no.y; ra.z = no.z; ra.w = wo.x;
    rb.x = wo.y; rb.y = wo.z;
}

// Camera::sample (camera.hpp:51-60) of sample `sl` of the lane's pixel
// ARGS_F32: the camera's constants come as the floats the host made of them (PathArgs::eye_f ...: scalar registers, re-loaded from the
// kernel's arguments where the compiler runs out of them).  Left to itself the compiler converts the double arguments with v_cvt_f32_f64,
// whose result is a VECTOR register: seventeen wave-uniform values lived through the whole kernel that way (eleven of them in scratch once
// the lockstep kernel was compiled for seven waves per SIMD).  Config 3's frame: 0.666 -> 0.659 ms, an albedo per shape 0.733 -> 0.725.  The
// regenerating form, whose scalar registers are all taken, keeps the conversions (0.353 against 0.360 ms with the arguments, which it
// re-loads in every iteration; from a table in LDS: 0.357, and the lockstep kernel 0.675).
#ifndef DRT_CAMERA_ARGS_F32
#define DRT_CAMERA_ARGS_F32 1
#endif
template <typename R, bool ARGS_F32 = (DRT_CAMERA_ARGS_F32 != 0)>
__device__ inline uint32_t path_camera(const PathArgs& a, const CameraLane<R>& cl, uint32_t gpix, uint32_t px, uint32_t py, uint32_t sl,
                                       typename Q4<R>::T& ra, typename Q2<R>::T& rb)
{
    const uint64_t path = (uint64_t)gpix * (uint64_t)a.spp + (uint64_t)(a.s0 + sl);
    const uint32_t key = (uint32_t)path;
    if (sizeof(R) == 4) {
        const float cs_step = ARGS_F32 ? a.cs_step_f : (float)(2. * a.aspect * a.tan_half * a.inv_W);
        const float ct_step = ARGS_F32 ? a.ct_step_f : (float)(2. * a.tan_half * a.inv_H);
        const V3<float> eye = ARGS_F32 ? mk<float>(a.eye_f[0], a.eye_f[1], a.eye_f[2]) : mk<float>((float)a.eye[0], (float)a.eye[1], (float)a.eye[2]);
        const V3<float> fw = ARGS_F32 ? mk<float>(a.fwd_f[0], a.fwd_f[1], a.fwd_f[2]) : mk<float>((float)a.fwd[0], (float)a.fwd[1], (float)a.fwd[2]);
        const V3<float> rt = ARGS_F32 ? mk<float>(a.right_f[0], a.right_f[1], a.right_f[2]) : mk<float>((float)a.right[0], (float)a.right[1], (float)a.right[2]);
        const V3<float> upv = ARGS_F32 ? mk<float>(a.up_f[0], a.up_f[1], a.up_f[2]) : mk<float>((float)a.up[0], (float)a.up[1], (float)a.up[2]);
        const float cs = fmaf(u01(0.f, rng_draw(a.rng_stream, key, 0)), cs_step, (float)cl.cs0);
        const float ct = fmaf(u01(0.f, rng_draw(a.rng_stream, key, 1)), ct_step, (float)cl.ct0);
        const V3<float> dir = mk<float>(fw.x + cs * rt.x - ct * upv.x, fw.y + cs * rt.y - ct * upv.y, fw.z + cs * rt.z - ct * upv.z);
        const V3<float> dn = normalize(dir);
        ra.x = (R)eye.x; ra.y = (R)eye.y; ra.z = (R)eye.z; ra.w = (R)dn.x;
        rb.x = (R)dn.y; rb.y = (R)dn.z;
    } else {                                          // f64 verification mode: the reference's own sequence, in double
        const double u1 = (double)rng_draw(a.rng_stream, key, 0) / DRT_RAND_MAX_D;
        const double u2 = (double)rng_draw(a.rng_stream, key, 1) / DRT_RAND_MAX_D;
        const double s = ((double)px + u1) / (double)a.W;
        const double t = ((double)py + u2) / (double)a.H;
        const double cs = (2. * s - 1.) * a.aspect * a.tan_half;
        const double ct = (2. * t - 1.) * a.tan_half;
        double dx = a.fwd[0] + cs * a.right[0] - ct * a.up[0];
        double dy = a.fwd[1] + cs * a.right[1] - ct * a.up[1];
        double dz = a.fwd[2] + cs * a.right[2] - ct * a.up[2];
        const double inv = 1.0 / sqrt(dx * dx + dy * dy + dz * dz);
        ra.x = (R)a.eye[0]; ra.y = (R)a.eye[1]; ra.z = (R)a.eye[2]; ra.w = (R)(dx * inv);
        rb.x = (R)(dy * inv); rb.y = (R)(dz * inv);
    }
    return key;
}
// ... of sample `sl` of pixel `gpix`, for a lane that is not bound to one pixel (k_path's regenerating form): `cl` holds that
// pixel's constants; the f64 mode needs the pixel's coordinates too
template <typename R>
__device__ inline uint32_t path_camera(const PathArgs& a, const CameraLane<R>& cl, uint32_t gpix, uint32_t sl,
                                       typename Q4<R>::T& ra, typename Q2<R>::T& rb)
{
    uint32_t px = 0, py = 0;
    if (sizeof(R) != 4) {
        py = gpix / (uint32_t)a.W;
        px = gpix - py * (uint32_t)a.W;
    }
    return path_camera<R, false>(a, cl, gpix, px, py, sl, ra, rb);
}

// blocks per CU (= waves per SIMD) a k_path instantiation is compiled for (its register budget); the knobs are above.
// A kernel hiprtc makes for a scene of MANY shapes (their records in scalar registers, their tests unrolled) needs the registers: rooms of
// 15 / 31 shapes, 4 parameters, ms per launch on config 3's frame at six waves with converted constants / seven / six with the constants as
// arguments / seven with them: 0.858 / 0.849 / 0.847 / 1.039 and 2.21 / 2.50 / 2.37 / 2.61 -- so the seventh wave and the float arguments
// (path_camera) are for kernels of up to DRT_LEAN_MAX_SHAPES compiled-in shapes (the reference's scene has 9) and for the run-time program.
// Kernels that carry caller-defined kinds (DRT_USER_SHAPES: the caller's own intersect / normal / BxDF bodies inlined) keep the wider budgets too:
// a power-cosine lobe from source lost 5-8 % at six waves (1.84 -> 1.93 ms, with a disc 1.94 -> 2.09).
#ifndef DRT_LEAN_MAX_SHAPES
#ifdef DRT_USER_SHAPES
#define DRT_LEAN_MAX_SHAPES (-1)
#else
#define DRT_LEAN_MAX_SHAPES 10
#endif
#endif
template <size_t RB, bool SPEC, int NP, int NSG, bool REGEN>
constexpr int path_min_blocks()
{
    if (RB == 4 && NP <= 4) {
        if (!REGEN)
            return SPEC ? ((NP == DRT_NP_ANY || NSG > DRT_LEAN_MAX_SHAPES) ? 5 : DRT_LOCKSTEP_SPEC_MIN_BLOCKS)
                        : (NP == DRT_NP_ANY ? DRT_LOCKSTEP_GEN_MIN_BLOCKS : (NSG <= DRT_LEAN_MAX_SHAPES ? DRT_LOCKSTEP_MIN_BLOCKS : DRT_LOCKSTEP_GEN_MIN_BLOCKS));
        return (NP == DRT_NP_ANY && NSG == 0) ? 4             // (general form + the kind-sorted program: 34 KB of LDS)
                                              : (SPEC ? 5 : DRT_REGEN_MIN_BLOCKS);
    }
    return (RB == 8 && NP <= 4 && !REGEN && !SPEC) ? DRT_F64_MIN_BLOCKS : 1;
}

// ---- the kernel -----------------------------------------------------------------------------------
// The bounce loop is written WITHOUT per-lane branches: every lane of the wave executes every bounce of the sample --
// a lane whose path has ended keeps tracing a stale ray whose results are never used (`live` guards every
// accumulation, its T and dT are frozen by selects) -- because that is what the SIMD does anyway, and straight-line
// code spares the exec-mask bookkeeping, the register copies at the joins and the waits in front of them.
// REGEN = false: all lanes of a wave trace their pixel's sample s at the same time and stand at the same depth (scalar
// depth bookkeeping; lanes whose path ended idle to the end of the sample: fine when paths end at a fixed depth).
// REGEN = true: every lane is on its own -- when its path ends (the roulette of pathtracer.hpp:128 ends paths at any
// depth) it starts its next sample at once, so no lane waits for the longest path of the wave: per-lane depth
// bookkeeping, the camera code runs whenever some lane starts over, the light's emission is added when the lane's
// path ends.  ~35 % more instructions per bounce, but roulette-terminated renders (the reference's defaults, -b 1
// -p 0.5: 2.5 vertices per path on average, some paths 20) keep their lanes busy.
// (waves per SIMD by form: path_min_blocks above)
template <typename R, bool SPEC, int NP, int NC, typename SG, bool REGEN = false, bool LOSS = false>
__global__ void __launch_bounds__(DRT_BLOCK, (path_min_blocks<sizeof(R), SPEC, NP, SG::n, REGEN>()))
k_path(PathArgs a, const DevScene<R>* __restrict__ sc, const R* __restrict__ params, const float* __restrict__ adjoint,
       double* __restrict__ gpart, double* __restrict__ fpart, uint32_t* __restrict__ counts,
       unsigned long long* __restrict__ total, double* __restrict__ gimg_part)
{
    if (total && blockIdx.x == 0 && threadIdx.x < 8)
        total[threadIdx.x] = 0;                           // (the finishing kernel behind this launch adds into them)
    typedef typename Q4<R>::T R4;
    typedef typename Q2<R>::T R2;
    __shared__ PathSceneLds<R> lds;
    __shared__ double s_red[REGEN ? 1 : DRT_BLOCK / DRT_WAVE][DRT_FAST_PARAMS * 3];   // (REGEN: the block's gradient partials reuse the pixel sums' table)
    __shared__ TangentLds<R> s_tl;
    constexpr bool GEN = NP == DRT_NP_ANY;                // any number of parameters: history + per-wave tables (see Tangents<R, DRT_NP_ANY>)
    __shared__ typename PickT<GEN, GenBlock<R>, NoLds>::T s_gen;
    extern __shared__ uint32_t s_hist[];                  // GEN: [a.hist_lds][DRT_BLOCK] history words
    if constexpr (GEN)
        gen_zero(s_gen);
    stage_path_scene(lds, sc, params);
    const TangentLds<R>& tl = s_tl;
    if (NC > 0 && !GEN)
        stage_tangents(s_tl, lds);

    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t w = grid_wave();                       // wave of the grid = group + n_groups * range
    const uint32_t range = w / a.n_groups, group = w - range * a.n_groups;
    const uint32_t lp = group * DRT_WAVE + lane;          // batch-local pixel of this lane
    const bool have = range < a.n_ranges && lp < a.Pb;
    const uint32_t s_begin = range * a.spr;
    const uint32_t s_end = s_begin + a.spr < a.Sb ? s_begin + a.spr : a.Sb;

    Tangents<R, NP, NC> tg;
    __shared__ R s_acc[NP > 0 ? NP * 3 : 1][DRT_BLOCK];
    tg.acc = &s_acc[0][threadIdx.x];
    if constexpr (GEN)
        gen_begin(s_gen, lds, sc, a, s_hist, reinterpret_cast<uint32_t*>(a.hist_ovf), tg);
    else {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            tg.acc_set(p, mk<R>(R(0), R(0), R(0)));
    }
    double fx = 0, fy = 0, fz = 0;                        // radiance sum of this lane's pixel over the range
    uint32_t n_seg = 0, n_capped = 0;                     // wave-uniform counters

    uint32_t gpix = 0, px = 0, py = 0;
    V3<R> g = mk<R>(R(1), R(1), R(1));                    // render.cpp:80: radiance.backward(Vec3(1))
    if (have) {
        gpix = path_global_pixel(a, a.p0 + lp);
        py = gpix / (uint32_t)a.W;
        px = gpix - py * (uint32_t)a.W;
        if (NP != 0 && adjoint)
            g = mk<R>((R)adjoint[(size_t)gpix * 3], (R)adjoint[(size_t)gpix * 3 + 1], (R)adjoint[(size_t)gpix * 3 + 2]);
    }
    // f32: the pixel's corner in double ONCE per lane; a sample then only adds its jitter (camera.hpp:53-58 in the form
    // cs = cs0 + u1 (2 aspect tan / W)): the sample's position inside the pixel is exact to ~2e-5 of a pixel and the
    // direction to 1 ulp of f32 -- the resolution the ray has anyway once it is stored in f32
    CameraLane<R> cl;
    cl.cs0 = (R)((2. * (double)px * a.inv_W - 1.) * a.aspect * a.tan_half);
    cl.ct0 = (R)((2. * (double)py * a.inv_H - 1.) * a.tan_half);
    constexpr bool ARGS_F32 = sizeof(R) == 4 && !REGEN && DRT_CAMERA_ARGS_F32 != 0 && SG::n <= DRT_LEAN_MAX_SHAPES;   // (see path_camera)
    const R pk_rr = ARGS_F32 ? (R)a.p_rr_f : (R)a.p_rr, inv_p_rr = ARGS_F32 ? (R)a.inv_p_rr_f : (R)a.inv_p_rr;
    ProgRecs<SG::n, R> recs;
    // (the kind-sorted program's 1.3 KB only where it runs: with the kinds compiled in they are what stands between the
    //  regenerating kernel and a sixth block per CU)
    __shared__ typename PickT<(SG::n == 0), ProgLds, NoLds>::T s_prog;
    recs.lds = reinterpret_cast<const ProgLds*>(&s_prog);
    if (SG::n > 0)
        recs.template load<SG>(sc);
    if constexpr (sizeof(R) == 4 && SG::n == 0) {
        const DevScene<float>* scf = reinterpret_cast<const DevScene<float>*>(sc);
        if (threadIdx.x < DRT_PROG_SORTED_MAX) {
            s_prog.rec[threadIdx.x] = *reinterpret_cast<const float4*>(scf->sorted[threadIdx.x]);
            s_prog.shape[threadIdx.x] = scf->sorted_shape[threadIdx.x];
        }
        if (threadIdx.x < 8)
            s_prog.kind_begin[threadIdx.x] = scf->kind_begin[threadIdx.x];
        __syncthreads();
    }

    if (range < a.n_ranges && !REGEN) {
    for (uint32_t sl = s_begin; sl < s_end; ++sl) {
        R4 ra;
        R2 rb;
        const uint32_t key = path_camera<R, ARGS_F32>(a, cl, gpix, px, py, sl, ra, rb);
        // pathtracer.hpp:128 at depth 0
        bool live = have && a.depth_cap > 0 && !(a.min_bounces <= 0 && rng_draw(a.rng_stream, key, 2) < a.rr_threshold);
        V3<R> T = mk<R>(R(1), R(1), R(1)), L = mk<R>(R(0), R(0), R(0));
        uint32_t end_ids = DRT_ID_NONE;                   // emission parameter of the light the path ended on
        R end_inv_pk = R(1);
        if (NC > 0 || GEN)
            tg.new_path();
        for (int kk = 0; kk < a.depth_cap; ++kk) {
            const uint32_t n_live = (uint32_t)__popcll(wave_ballot(live));
            if (n_live == 0)
                break;
            n_seg += n_live;
            const R pk = kk >= a.min_bounces ? pk_rr : R(1);                  // pathtracer.hpp:130
            const R inv_pk = kk >= a.min_bounces ? inv_p_rr : R(1);
            const uint32_t n_theta = draw_offset(kk, 0, a.min_bounces) + camera_draw_base(a.min_bounces);
            const bool next_rr = (kk + 1) >= a.min_bounces;
            const bool next_cap = (kk + 1) >= a.depth_cap;
            bool alive, capped, on_light;
            uint32_t light;
            path_bounce<R, SPEC, NP, NC, SG>(a, lds, tl, sc, params, recs, key, pk, inv_pk, n_theta, next_rr, next_cap, live, g,
                                                    ra, rb, T, L, tg, alive, capped, on_light, light, nullptr, next_cap);
            // A light without a BxDF ends the path: T and dT stay as they are in this lane, so its emission is added
            // ONCE PER SAMPLE, after the bounce loop, for all lanes together -- not here, where every bounce a few lanes
            // of the wave would drag the other sixty through it.
            end_ids = on_light ? light : end_ids;
            end_inv_pk = on_light ? inv_pk : end_inv_pk;
            if (next_cap && !a.cap_is_roulette)
                n_capped += (uint32_t)__popcll(wave_ballot(capped));
            live = alive;
        }
        if (wave_any(end_ids != DRT_ID_NONE)) {
            if (end_ids != DRT_ID_NONE)
                add_emission<R, NP, NC, LOSS>(lds, tl, params, end_ids, end_inv_pk, T, g, L, tg);
        }
        fx += (double)L.x; fy += (double)L.y; fz += (double)L.z;
    }
    }
    // ---- REGEN: a wave owns 64 pixels x its sample range, and its LANES are not bound to pixels: a lane whose path has ended
    // takes the NEXT camera sample of the wave (item n = pixel n % 64, sample n / 64: wave-uniform counter + the lane's rank
    // among the takers), so all 64 lanes stay busy until the wave's samples run out -- with a lane = a pixel the wave lasted as
    // long as its unluckiest lane's 32 paths (82 bounces where the mean lane has 62 under the reference's -b 1 -p 0.5: a
    // quarter of the wave-time idle).  What belongs to a PIXEL -- camera constants, radiance sum -- therefore
    // lives in per-wave tables in LDS (slot = the pixel's lane of the static layout, which also writes the sums out at the
    // end); radiance is added there with ds_add_f64.  (The sums of f32 values in f64 are exact unless a pixel's samples span
    // more than 2^24 in magnitude, so their order does not show; gradient sums are per lane and summed over lanes anyway.)
    __shared__ R s_pcs[REGEN ? DRT_BLOCK : 1], s_pct[REGEN ? DRT_BLOCK : 1];
    __shared__ uint32_t s_pgpix[REGEN ? DRT_BLOCK : 1];
    __shared__ double s_film[REGEN ? 3 : 1][REGEN ? DRT_BLOCK : 1];
    __shared__ uint32_t s_ih[REGEN ? DRT_DRAW_TABLE : 1];      // h(n) of every draw index a path of <= DRT_MAX_DEPTH vertices can reach
    if (REGEN) {
        for (uint32_t n = threadIdx.x; n < DRT_DRAW_TABLE; n += DRT_BLOCK)
            s_ih[REGEN ? n : 0] = drt_rng_index_hash(a.rng_stream, n);
        s_pcs[threadIdx.x] = cl.cs0;
        s_pct[threadIdx.x] = cl.ct0;
        s_pgpix[threadIdx.x] = have ? gpix : 0xFFFFFFFFu;       // (no such pixel: its samples are skipped)
        s_film[0][threadIdx.x] = 0.0; s_film[REGEN ? 1 : 0][threadIdx.x] = 0.0; s_film[REGEN ? 2 : 0][threadIdx.x] = 0.0;
        __syncthreads();
    }
    if (range < a.n_ranges && REGEN) {
        const uint32_t wbase = threadIdx.x & ~(uint32_t)(DRT_WAVE - 1);      // this wave's first slot in the pixel tables
        const uint32_t n_items = DRT_WAVE * (s_end - s_begin);
        uint32_t n_next = 0;                              // wave-uniform: the next camera sample of the wave to hand out
        uint32_t key = 0, pix = wbase, pgpix = 0;         // RNG key, pixel slot and pixel of the lane's current path
        int kk = 0;                                       // depth of the lane's current path
        bool live = false;
        R4 ra;
        R2 rb;
        ra.x = ra.y = ra.z = ra.w = R(0);
        rb.x = rb.y = R(0);
        V3<R> T = mk<R>(R(1), R(1), R(1)), L = mk<R>(R(0), R(0), R(0));
        if (NC > 0 || GEN)
            tg.new_path();
        const int first_rr = a.min_bounces > 1 ? a.min_bounces : 1;
        for (;;) {
            // ---- lanes without a path take the wave's next samples -- once enough of them wait (the whole wave walks
            // through the camera code), or as many as still run
            const uint64_t idle_mask = wave_ballot(!live);
            const uint32_t n_idle = (uint32_t)__popcll(idle_mask), n_run = DRT_WAVE - n_idle;
            if (n_next < n_items && (n_idle >= a.regen_min || n_idle >= n_run)) {
                const uint32_t n = n_next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                if (!live && n < n_items) {
                    pix = wbase + (n & (DRT_WAVE - 1));
                    const uint32_t gp = s_pgpix[pix];
                    if (gp != 0xFFFFFFFFu) {
                        pgpix = gp;
                        CameraLane<R> pc;
                        pc.cs0 = s_pcs[pix];
                        pc.ct0 = s_pct[pix];
                        key = path_camera<R>(a, pc, gp, s_begin + (n >> 6), ra, rb);
                        kk = 0;
                        live = a.depth_cap > 0 && !(a.min_bounces <= 0 && rng_draw(a.rng_stream, key, 2) < a.rr_threshold);
                        T = mk<R>(R(1), R(1), R(1));
                        L = mk<R>(R(0), R(0), R(0));
                        if (NC > 0 || GEN)
                            tg.new_path();
                    }
                }
                n_next += n_idle < n_items - n_next ? n_idle : n_items - n_next;
            }
            const uint32_t n_live = (uint32_t)__popcll(wave_ballot(live));
            if (n_live == 0) {
                if (n_next >= n_items)
                    break;                                // the wave is through its samples
                continue;                                 // (all fresh paths were absorbed at depth 0, or belong to no pixel)
            }
            n_seg += n_live;
            // ---- one bounce, every lane at its own depth
            const bool rr_here = kk >= a.min_bounces;
            const R pk = rr_here ? pk_rr : R(1);                              // pathtracer.hpp:130
            const R inv_pk = rr_here ? inv_p_rr : R(1);
            const int rr_draws = kk - first_rr + 1;                           // roulette draws at depths 1 .. kk (draw_offset)
            const uint32_t n_theta = 2u * (uint32_t)kk + (uint32_t)(rr_draws > 0 ? rr_draws : 0) + camera_draw_base(a.min_bounces);
            const bool next_rr = (kk + 1) >= a.min_bounces;
            const bool next_cap = (kk + 1) >= a.depth_cap;
            bool alive, capped, on_light;
            uint32_t light;
            // the seed of the path's pixel (render.cpp:80: all ones; else the caller's image): read where a path meets a light
            V3<R> gp3 = mk<R>(R(1), R(1), R(1));
            const float* seed_px = (NP != 0 && adjoint) ? adjoint + (size_t)pgpix * 3 : (const float*)nullptr;
            path_bounce<R, SPEC, NP, NC, SG>(a, lds, tl, sc, params, recs, key, pk, inv_pk, n_theta, next_rr, next_cap, live, gp3,
                                                    ra, rb, T, L, tg, alive, capped, on_light, light, nullptr, false, seed_px, s_ih);
            if (!a.cap_is_roulette)
                n_capped += (uint32_t)__popcll(wave_ballot(capped));
            // ---- paths that ended here hand their radiance to their pixel
            const bool ended = live && !alive;
            if (wave_any(ended)) {
                if (ended) {
                    if (on_light) {
                        if (seed_px)
                            gp3 = mk<R>((R)seed_px[0], (R)seed_px[1], (R)seed_px[2]);
                        add_emission<R, NP, NC, LOSS>(lds, tl, params, light, inv_pk, T, gp3, L, tg);
                    }
                    if (L.x != R(0) || L.y != R(0) || L.z != R(0)) {
                        atomicAdd(&s_film[0][pix], (double)L.x);
                        atomicAdd(&s_film[REGEN ? 1 : 0][pix], (double)L.y);
                        atomicAdd(&s_film[REGEN ? 2 : 0][pix], (double)L.z);
                    }
                }
            }
            live = alive;
            ++kk;
        }
    }
    if (REGEN) {
        __syncthreads();                                  // (every wave's adds have landed)
        fx = s_film[0][threadIdx.x]; fy = s_film[REGEN ? 1 : 0][threadIdx.x]; fz = s_film[REGEN ? 2 : 0][threadIdx.x];
    }

    if (range < a.n_ranges) {
        if (fpart && have) {
            double* f = fpart + ((size_t)range * 3) * a.Pb + lp;       // [range][channel][pixel]: coalesced
            f[0] = fx; f[(size_t)a.Pb] = fy; f[(size_t)a.Pb * 2] = fz;
        }
        if constexpr (GEN && NC == 1) if (gimg_part && have) {
            // gradient image, general form: the lane's own adds to the row of the image's parameter
            double* f = gimg_part + ((size_t)range * 3) * a.Pb + lp;
            f[0] = (double)tg.gsum.x; f[(size_t)a.Pb] = (double)tg.gsum.y; f[(size_t)a.Pb * 2] = (double)tg.gsum.z;
        }
        if constexpr (NP > 0) if (gimg_part && have) {
            // gradient image (README.md:142-145): a lane IS a pixel, its gradient sum of one parameter over the samples of
            // this range is that pixel's share -- same layout as the radiance partials, same finishing kernels
            const V3<R> v = tg.acc_get(a.gimg_param > 0 && a.gimg_param < NP ? a.gimg_param : 0);
            double* f = gimg_part + ((size_t)range * 3) * a.Pb + lp;
            f[0] = (double)v.x; f[(size_t)a.Pb] = (double)v.y; f[(size_t)a.Pb * 2] = (double)v.z;
        }
        if (lane == 0) {
            counts[w] = n_seg;
            counts[(size_t)a.n_groups * a.n_ranges + w] = n_capped;
        }
    }
    if constexpr (GEN)
        gen_finish(s_gen, a, gpart);
    else
    if (NP > 0) {
        // block reduction in fp64: thread -> wave (shuffles) -> block (LDS), fixed order; K7 adds the blocks
        const int wv = threadIdx.x / DRT_WAVE;
        double (*red)[DRT_FAST_PARAMS * 3] = s_red;
        if (REGEN) {
            __syncthreads();                               // (every thread has read its pixel's sums out of s_film)
            red = reinterpret_cast<double (*)[DRT_FAST_PARAMS * 3]>(&s_film[0][0]);
        }
#pragma unroll
        for (int r = 0; r < NP * 3; ++r) {
            double v = (double)tg.acc[r * DRT_BLOCK];
#pragma unroll
            for (int o2 = DRT_WAVE / 2; o2 > 0; o2 >>= 1)
                v += __shfl_down(v, o2);
            if (lane == 0)
                red[wv][r] = v;
        }
        __syncthreads();
        if (threadIdx.x < DRT_FAST_PARAMS * 3) {
            double v = 0;
            if ((int)threadIdx.x < NP * 3)
                for (int ww = 0; ww < DRT_BLOCK / DRT_WAVE; ++ww)
                    v += red[ww][threadIdx.x];
            gpart[(size_t)blockIdx.x * (DRT_FAST_PARAMS * 3) + threadIdx.x] = v;
        }
    }
}

// ---- the unbiased integration operator (integrate.hpp:11-24, 39-52; README.md:104-136) in one launch ------------------
// The reference's IntegrateBackward does not reuse the forward samples: where a gradient arrives at a vertex it draws a
// FRESH direction, evaluates forward(sample) -- a whole new suffix path -- back-propagates grad / pdf through
// brdf * radiance * cos, and the recursion continues down the NEW path (O(depth^2) segments per camera sample).  Like
// k_path, a lane is a pixel and everything lives in registers:
//   forward walk from the eye (radiance -> the image; its first vertex starts the chain), then per chain vertex:
//   emission gradient, fresh direction, suffix walk (radiance L'), colour gradient += L' * g / pdf * cos * bs,
//   g *= f * cos / pdf, and the suffix's first vertex is the next chain vertex.
// All draws of a path come from ONE stream in the order the reference consumes them (a counter per lane): camera 2,
// then per vertex [roulette of its depth] theta phi -- including the roulette draw the reference spends AFTER a light
// without BxDF (its zero-direction continuation is traced "faithfully" before it misses; oracle/ref_harness.cpp).
// One walk = the bounce of k_path (path_bounce, forward-only) iterated while any lane of the wave still traces.
template <typename R, bool SPEC, typename SG>
__device__ inline void unbiased_walk(const PathArgs& a, const PathSceneLds<R>& lds, const TangentLds<R>& tl, const DevScene<R>* __restrict__ sc,
                                     const R* __restrict__ params, const ProgRecs<SG::n, R>& recs, uint32_t key,
                                     R pk_rr, R inv_p_rr, bool live, typename Q4<R>::T ra, typename Q2<R>::T rb, int kk,
                                     uint32_t& nd, uint32_t& n_seg, uint32_t& n_capped, V3<R>& L, PathVertex<R>& first, bool& any_vertex)
{
    V3<R> T = mk<R>(R(1), R(1), R(1));
    L = mk<R>(R(0), R(0), R(0));
    any_vertex = false;
    Tangents<R, 0, 0> none;
    const V3<R> g1 = mk<R>(R(1), R(1), R(1));
    for (;;) {
        const uint32_t n_live = (uint32_t)__popcll(wave_ballot(live));
        if (n_live == 0)
            break;
        n_seg += n_live;
        const bool rr_here = kk >= a.min_bounces;
        const R pk = rr_here ? pk_rr : R(1), inv_pk = rr_here ? inv_p_rr : R(1);
        const bool next_rr = (kk + 1) >= a.min_bounces, next_cap = (kk + 1) >= a.depth_cap;
        bool alive, capped, on_light;
        uint32_t light;
        PathVertex<R> v;
        path_bounce<R, SPEC, 0, 0, SG>(a, lds, tl, sc, params, recs, key, pk, inv_pk, nd, next_rr, next_cap, live, g1, ra, rb, T, L,
                                              none, alive, capped, on_light, light, &v);
        // the roulette of the next depth is drawn unless a user cap ends the path first (pathtracer.hpp:128 behind the cap test)
        const uint32_t rr_drawn = (next_rr && (!next_cap || a.cap_draws)) ? 1u : 0u;
        if (live) {
            if (!any_vertex && v.hit) {
                first = v;
                any_vertex = true;
            }
            nd += v.scattered ? 2u + rr_drawn : (v.hit ? rr_drawn : 0u);
        }
        if (wave_any(live && on_light)) {
            if (live && on_light)
                add_emission<R, 0, 0>(lds, tl, params, light, inv_pk, T, g1, L, none);
        }
        if (!a.cap_is_roulette)
            n_capped += (uint32_t)__popcll(wave_ballot(live && capped));
        live = alive;
        ++kk;
    }
}

#ifndef DRT_UNBIASED_MIN_BLOCKS
#define DRT_UNBIASED_MIN_BLOCKS 4    // blocks per CU the f32 k_path_unbiased is compiled for: 128 registers + 96 B of scratch per lane.  Left to itself the
                                     // compiler takes 153-173 registers (three / two waves per SIMD).  Config 3's frame, ms per launch, builds alternating:
                                     // diffuse 5.47 -> 4.77 at four (five: 4.93, six: 5.56), with the glossy sphere 9.41 -> 6.65 (7.09, 7.98)
#endif
template <typename R, bool SPEC, int NP, typename SG>
__global__ void __launch_bounds__(DRT_BLOCK, (sizeof(R) == 4 ? DRT_UNBIASED_MIN_BLOCKS : 1))
k_path_unbiased(PathArgs a, const DevScene<R>* __restrict__ sc, const R* __restrict__ params, const float* __restrict__ adjoint,
                double* __restrict__ gpart, double* __restrict__ fpart, uint32_t* __restrict__ counts,
                unsigned long long* __restrict__ total)
{
    if (total && blockIdx.x == 0 && threadIdx.x < 8)
        total[threadIdx.x] = 0;
    typedef typename Q4<R>::T R4;
    typedef typename Q2<R>::T R2;
    __shared__ PathSceneLds<R> lds;
    __shared__ double s_red[DRT_BLOCK / DRT_WAVE][DRT_FAST_PARAMS * 3];
    // NP = DRT_NP_ANY: any number of parameters -- the chain's two adds per vertex (the light's colour, the BxDF's colour) go
    // to the wave's gradient table in LDS (Tangents<R, DRT_NP_ANY>; no vertex history: the chain IS the reverse sweep)
    constexpr bool GEN = NP == DRT_NP_ANY;
    __shared__ typename PickT<GEN, GenBlock<R>, NoLds>::T s_gen;
    if constexpr (GEN)
        gen_zero(s_gen);
    stage_path_scene(lds, sc, params);
    const TangentLds<R>& tl = *reinterpret_cast<const TangentLds<R>*>(&lds);   // (forward-only walks never touch it)
    Tangents<R, GEN ? DRT_NP_ANY : 0, 0> gt;
    if constexpr (GEN)
        gen_begin(s_gen, lds, sc, a, (uint32_t*)nullptr, (uint32_t*)nullptr, gt);

    const uint32_t lane = threadIdx.x & (DRT_WAVE - 1);
    const uint32_t w = grid_wave();
    const uint32_t range = w / a.n_groups, group = w - range * a.n_groups;
    const uint32_t lp = group * DRT_WAVE + lane;
    const bool have = range < a.n_ranges && lp < a.Pb;
    const uint32_t s_begin = range * a.spr;
    const uint32_t s_end = s_begin + a.spr < a.Sb ? s_begin + a.spr : a.Sb;

    V3<R> acc[NP > 0 ? NP : 1];
#pragma unroll
    for (int p = 0; p < NP; ++p)
        acc[p] = mk<R>(R(0), R(0), R(0));
    double fx = 0, fy = 0, fz = 0;
    uint32_t n_seg = 0, n_capped = 0;
    uint32_t gpix = 0, px = 0, py = 0;
    V3<R> g0 = mk<R>(R(1), R(1), R(1));                   // render.cpp:80: radiance.backward(Vec3(1))
    if (have) {
        gpix = path_global_pixel(a, a.p0 + lp);
        py = gpix / (uint32_t)a.W;
        px = gpix - py * (uint32_t)a.W;
        if (adjoint)
            g0 = mk<R>((R)adjoint[(size_t)gpix * 3], (R)adjoint[(size_t)gpix * 3 + 1], (R)adjoint[(size_t)gpix * 3 + 2]);
    }
    CameraLane<R> cl;
    cl.cs0 = (R)((2. * (double)px * a.inv_W - 1.) * a.aspect * a.tan_half);
    cl.ct0 = (R)((2. * (double)py * a.inv_H - 1.) * a.tan_half);
    constexpr bool ARGS_F32 = sizeof(R) == 4 && DRT_CAMERA_ARGS_F32 != 0 && SG::n <= DRT_LEAN_MAX_SHAPES;   // (see path_camera)
    const R pk_rr = ARGS_F32 ? (R)a.p_rr_f : (R)a.p_rr, inv_p_rr = ARGS_F32 ? (R)a.inv_p_rr_f : (R)a.inv_p_rr;
    ProgRecs<SG::n, R> recs;
    __shared__ ProgLds s_prog;
    recs.lds = &s_prog;
    if (SG::n > 0)
        recs.template load<SG>(sc);
    if (sizeof(R) == 4 && SG::n == 0) {
        const DevScene<float>* scf = reinterpret_cast<const DevScene<float>*>(sc);
        if (threadIdx.x < DRT_PROG_SORTED_MAX) {
            s_prog.rec[threadIdx.x] = *reinterpret_cast<const float4*>(scf->sorted[threadIdx.x]);
            s_prog.shape[threadIdx.x] = scf->sorted_shape[threadIdx.x];
        }
        if (threadIdx.x < 8)
            s_prog.kind_begin[threadIdx.x] = scf->kind_begin[threadIdx.x];
        __syncthreads();
    }
    // is the roulette of depth d drawn at all?  (not behind a user cap: the reference tests the cap first)
    auto rr_drawn = [&](int d) { return d >= a.min_bounces && (d < a.depth_cap || a.cap_draws != 0); };

    if (range < a.n_ranges) {
        for (uint32_t sl = s_begin; sl < s_end; ++sl) {
            R4 ra;
            R2 rb;
            const uint32_t key = path_camera<R, ARGS_F32>(a, cl, gpix, px, py, sl, ra, rb);
            uint32_t nd = 2;                                  // the camera's two draws
            bool live = have && a.depth_cap > 0;
            if (rr_drawn(0)) {                                // pathtracer.hpp:128 at depth 0
                live = live && !(rng_draw(a.rng_stream, key, nd) < a.rr_threshold);
                ++nd;
            }
            V3<R> L0;
            PathVertex<R> cur;
            bool in_chain;
            unbiased_walk<R, SPEC, SG>(a, lds, tl, sc, params, recs, key, pk_rr, inv_p_rr, live, ra, rb, 0, nd, n_seg, n_capped, L0,
                                              cur, in_chain);
            fx += (double)L0.x; fy += (double)L0.y; fz += (double)L0.z;
            in_chain = in_chain && have;
            int cdepth = 0;
            V3<R> g = g0;
            while (wave_any(in_chain)) {
                // ---- at the chain vertex: emission gradient, fresh direction (integrate.hpp:13-16)
                V3<R> gq = mk<R>(R(0), R(0), R(0)), fcol = gq;
                R bs = R(0);
                bool go = false;
                R4 sa = ra;
                R2 sb = rb;
                uint32_t cid = DRT_ID_NONE;
                if (in_chain) {
                    const R inv_pk = cdepth >= a.min_bounces ? inv_p_rr : R(1);
                    const V3<R> g1 = g * inv_pk;                  // "/ p" of trace()
                    const uint32_t eid = cur.ids >> 16;
                    cid = cur.ids & 0xFFFFu;
                    if constexpr (GEN) {
                        if (eid < DRT_PATH_LDS_PARAMS) {
                            const uint32_t row = pid_unpack(gt.gl->invc[eid][3]) & 0xFFFFu;
                            if (row != DRT_SLOT_NONE)
                                gt.add(row, g1);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const bool own = eid == (uint32_t)p;
                        acc[p] = mk<R>(acc[p].x + (own ? g1.x : R(0)), acc[p].y + (own ? g1.y : R(0)), acc[p].z + (own ? g1.z : R(0)));
                    }
                    if (cid == DRT_ID_NONE) {
                        in_chain = false;                         // no BxDF: f = 0, nothing below
                    } else {
                        const DevMaterial<R>& m = lds.sc.materials[cur.material];
                        V3<R> wo;
                        R q;
                        sample_bxdf<R, SPEC>(m, cur.nrm, cur.d, rng_draw(a.rng_stream, key, nd), rng_draw(a.rng_stream, key, nd + 1), wo, q, bs);
                        nd += 2;
                        const R c = dot(cur.nrm, wo);
                        gq = g1 * div_r(c, q);                    // grad / pdf (integrate.hpp:17), times cos
                        fcol = load_param<R, true>(lds, params, (int)cid) * bs;
                        go = (cdepth + 1) < a.depth_cap;
                        if (rr_drawn(cdepth + 1)) {
                            go = go && !(rng_draw(a.rng_stream, key, nd) < a.rr_threshold);
                            ++nd;
                        }
                        const V3<R> no = cur.P + wo * R(1e-3);
                        sa.x = no.x; sa.y = no.y; sa.z = no.z; sa.w = wo.x;
                        sb.x = wo.y; sb.y = wo.z;
                    }
                }
                // ---- forward(sample): the suffix from the fresh direction
                V3<R> Ls;
                PathVertex<R> nxt;
                bool any_vertex;
                unbiased_walk<R, SPEC, SG>(a, lds, tl, sc, params, recs, key, pk_rr, inv_p_rr, in_chain && go, sa, sb, cdepth + 1, nd,
                                                  n_seg, n_capped, Ls, nxt, any_vertex);
                // ---- gradients of this vertex; the chain moves on to the suffix's first vertex
                if (in_chain) {
                    const V3<R> df = Ls * gq * bs;                // MulBackward, brdf side
                    if constexpr (GEN) {
                        if (cid < DRT_PATH_LDS_PARAMS) {
                            const uint32_t row = pid_unpack(gt.gl->invc[cid][3]) & 0xFFFFu;
                            if (row != DRT_SLOT_NONE)
                                gt.add(row, df);
                        }
                    }
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const bool own = cid == (uint32_t)p;
                        acc[p] = mk<R>(acc[p].x + (own ? df.x : R(0)), acc[p].y + (own ? df.y : R(0)), acc[p].z + (own ? df.z : R(0)));
                    }
                    if (!any_vertex) {
                        in_chain = false;                         // the suffix is a constant
                    } else {
                        g = fcol * gq;                            // MulBackward, radiance side
                        cur = nxt;
                        ++cdepth;
                    }
                }
            }
        }
        if (fpart && have) {
            double* f = fpart + ((size_t)range * 3) * a.Pb + lp;
            f[0] = fx; f[(size_t)a.Pb] = fy; f[(size_t)a.Pb * 2] = fz;
        }
        if (lane == 0) {
            counts[w] = n_seg;
            counts[(size_t)a.n_groups * a.n_ranges + w] = n_capped;
        }
    }
    if constexpr (GEN)
        gen_finish(s_gen, a, gpart);
    else
    {   // block reduction in fp64 (as in k_path)
        const int wv = threadIdx.x / DRT_WAVE;
#pragma unroll
        for (int r = 0; r < NP * 3; ++r) {
            const V3<R> v3 = acc[r / 3];
            double v = (double)(r % 3 == 0 ? v3.x : (r % 3 == 1 ? v3.y : v3.z));
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

// film[p0 + j] += sum over the sample ranges of one batch, in range order (deterministic)
__global__ void __launch_bounds__(DRT_BLOCK)
k_film_parts(const double* __restrict__ fpart, uint32_t n_ranges, uint32_t Pb, uint32_t p0, double* __restrict__ film)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < Pb; j += stride) {
        double r = 0, g = 0, b = 0;
        for (uint32_t q = 0; q < n_ranges; ++q) {
            const double* f = fpart + ((size_t)q * 3) * Pb + j;
            r += f[0]; g += f[(size_t)Pb]; b += f[(size_t)Pb * 2];
        }
        double* f = film + (size_t)(p0 + j) * 3;
        f[0] += r; f[1] += g; f[2] += b;
    }
}

// ---- everything behind a k_path launch that covers the whole frame, in ONE launch --------------------------------
// (a single-batch render otherwise enqueues three memsets and four small kernels: film sums, resolve, K7, segment
// totals -- 30 us of launches around a 15 us k_path on a 128 x 128 x 4 frame).  Same sums in the same order as
// k_film_parts + k_resolve, k_gradreduce and k_sum_counts, written instead of added to zeroed buffers: bitwise the same.
//   blocks [0, film_blocks):                  out[pixel] = float(sum over the sample ranges, in range order, / spp)
//   blocks [film_blocks, + grad_words):       grad[p] = sum over k_path's blocks of gpart[block][p], fixed order (p >= n_rows: 0)
//   the rest:                                 total[0] += segments, total[3] += capped paths (integers; k_path zeroed them)
__global__ void __launch_bounds__(DRT_BLOCK)
k_path_finish(PathArgs a, const double* __restrict__ fpart, float* __restrict__ out, uint32_t film_blocks,
              const double* __restrict__ gpart, int n_blocks, int n_rows, int row_stride, double* __restrict__ grad, uint32_t grad_words,
              const uint32_t* __restrict__ counts, uint32_t n_waves, unsigned long long* __restrict__ total, uint32_t count_rows = 2,
              const unsigned short* __restrict__ slot_map = nullptr)
{
    // (slot_map, the general form of the path kernels: gpart's rows are table rows, DevScene::grad_slot maps parameters to them)
    // (count_rows = 3, k_path_mesh: a third word per wave, the rays its BVH walk took -> total[5])
    __shared__ double red[DRT_BLOCK];
    if (blockIdx.x < film_blocks) {
        const uint32_t stride = film_blocks * DRT_BLOCK;
        const double inv = 1.0 / (double)a.spp;
        for (uint32_t j = blockIdx.x * DRT_BLOCK + threadIdx.x; j < a.Pb; j += stride) {
            double r = 0, g = 0, b = 0;
            for (uint32_t q = 0; q < a.n_ranges; ++q) {
                const double* f = fpart + ((size_t)q * 3) * a.Pb + j;
                r += f[0]; g += f[(size_t)a.Pb]; b += f[(size_t)a.Pb * 2];
            }
            const uint32_t gp = path_global_pixel(a, a.p0 + j);
            // (one 12-byte store per lane: `out` may be host memory -- asynchronous host-buffer renders write the image
            //  straight into the pinned block -- and whole lines travel better than three strided dwords)
            drt_f3 px;
            px.x = (float)(r * inv); px.y = (float)(g * inv); px.z = (float)(b * inv);
            *reinterpret_cast<drt_f3_u*>(out + (size_t)gp * 3) = px;
        }
        return;
    }
    if (blockIdx.x < film_blocks + grad_words) {
        const int p = (int)(blockIdx.x - film_blocks);
        int src = p;
        if (slot_map) {
            const int slot = slot_map[p / 3];
            src = slot == (int)DRT_SLOT_NONE ? n_rows : slot * 3 + p % 3;
        }
        double v = 0;
        if (src < n_rows)
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
            grad[p] = red[0];
        return;
    }
    {
        const uint32_t nb = gridDim.x - film_blocks - grad_words, me = blockIdx.x - film_blocks - grad_words;
        unsigned long long seg = 0, cap = 0, wlk = 0;
        for (uint32_t i = me * DRT_BLOCK + threadIdx.x; i < n_waves; i += nb * DRT_BLOCK) {
            seg += counts[i];
            cap += counts[(size_t)n_waves + i];
            if (count_rows > 2)
                wlk += counts[(size_t)n_waves * 2 + i];
        }
        for (int off = DRT_WAVE / 2; off > 0; off >>= 1) {
            seg += __shfl_down(seg, off);
            cap += __shfl_down(cap, off);
            wlk += __shfl_down(wlk, off);
        }
        if ((threadIdx.x & (DRT_WAVE - 1)) == 0) {
            if (seg) atomicAdd(total, seg);
            if (cap) atomicAdd(total + 3, cap);
            if (wlk) atomicAdd(total + 5, wlk);
        }
    }
}
