// drt_render_impl.h -- one shard's render, enqueued: the k_path route (the whole path in one launch; its instantiation for
// the scene's shape kinds comes from the library, or from hiprtc: drt_jit.h) and the queue wavefront (K1-K7 over ray queues
// in HBM: meshes, more than eight parameters, one launch per bounce on request).  Replaces the pixel x sample loop of the
// reference's src/render.cpp:72-86.
#pragma once

namespace {

// event-bracketed launch bookkeeping (DRT_RENDER_TIMING)
int timing_begin(drt_hip_ctx* ctx, bool on, int kernel)
{
    if (!on)
        return DRT_OK;
    while (ctx->event_pool.size() < ctx->events_used + 2) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreate(&e));
        ctx->event_pool.push_back(e);
    }
    TimedLaunch t;
    t.kernel = kernel;
    t.e0 = ctx->event_pool[ctx->events_used++];
    t.e1 = ctx->event_pool[ctx->events_used++];
    HIPCHK(ctx, hipEventRecord(t.e0, ctx->stream));
    ctx->timed.push_back(t);
    return DRT_OK;
}

int timing_end(drt_hip_ctx* ctx, bool on)
{
    if (!on)
        return DRT_OK;
    HIPCHK(ctx, hipEventRecord(ctx->timed.back().e1, ctx->stream));
    return DRT_OK;
}

// a multiplier coprime to n (the walk's pull order is a multiplicative permutation of its list groups)
uint32_t coprime_multiplier(uint32_t n)
{
    if (n <= 2)
        return 1;
    uint32_t m = (uint32_t)(2654435761ull % n);
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
    while (m < 2 || gcd(m, n) != 1)
        m = m + 1 < n ? m + 1 : 2;
    return m;
}

int grid_for(const drt_hip_ctx* ctx, uint64_t work)
{
    uint64_t blocks = (work + DRT_BLOCK - 1) / DRT_BLOCK;
    uint64_t cap = (uint64_t)ctx->n_cu * 8;   // persistent grid: 8 x 256-thread blocks per CU
    if (blocks > cap)
        blocks = cap;
    if (blocks < 1)
        blocks = 1;
    return (int)blocks;
}

// Number of rays queued at one depth (sum over regions).  A host round trip: only used every few
// bounces when the depth cap is deep (roulette-terminated renders), to stop launching on empty queues.
int queue_length(drt_hip_ctx* ctx, const uint32_t* counts_row, uint32_t n_regions, unsigned long long* out)
{
    int rc;
    if ((rc = ensure(ctx, ctx->probe, 4 * sizeof(unsigned long long))) != DRT_OK) return rc;
    if (!ctx->h_probe)
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_probe, sizeof(unsigned long long)));
    HIPCHK(ctx, hipMemsetAsync(ctx->probe.p, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_sum_counts, dim3(16), dim3(DRT_BLOCK), 0, ctx->stream, counts_row, n_regions,
                       (unsigned long long*)ctx->probe.p, n_regions, 0ull, 0ull, 0xFFFFFFFFu);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_probe, ctx->probe.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *out = *ctx->h_probe;
    return DRT_OK;
}
// The instantiation `name_expr` of a kernel template of drt_path.h, compiled for this scene's KindSig by hiprtc and loaded
// on this context's device (drt_jit.h).  nullptr: it could not be made (ctx->jit_error says why; the caller renders with
// the kind-sorted program, same results).  wait = false: also nullptr while the compiler is still at it on its own thread --
// nothing is recorded then, the next frame asks again.
hipFunction_t jit_function(drt_hip_ctx* ctx, const std::string& name_expr, bool wait = true)
{
    auto it = ctx->jit_fn.find(name_expr);
    if (it != ctx->jit_fn.end())
        return it->second;
    const auto t0 = std::chrono::steady_clock::now();
    hipFunction_t fn = nullptr;
    const drt_jit::EntryPtr pe = wait ? drt_jit::compile(ctx->arch, name_expr, ctx->user_header) : drt_jit::poll(ctx->arch, name_expr, ctx->user_header);
    if (!pe)
        return nullptr;
    const drt_jit::Code& c = pe->code;
    if (!c.ok) {
        ctx->jit_error = c.log;
    } else {
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, c.bin.data());
        if (e == hipSuccess) {
            ctx->jit_modules.push_back(mod);
            e = hipModuleGetFunction(&fn, mod, c.lowered.c_str());
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fn = nullptr;
            ctx->jit_error = std::string("loading ") + name_expr + ": " + hipGetErrorString(e);
        }
    }
    if (!fn && tuning().jit_verbose)
        fprintf(stderr, "[drt_hip] specialisation failed: %s\n", ctx->jit_error.c_str());
    ctx->jit_fn[name_expr] = fn;
    // (what the kernel cost: the compile, on whichever thread it ran, and this context's load)
    ctx->jit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() + (wait ? 0. : c.ms);
    return fn;
}
// A compile costs ~0.5 s of host time and buys ~25 % of the kind-sorted program's time: it pays once the scene has rendered
// a few seconds' worth of frames.  2^31 path-bounces are ~20 ms of rendering: small test frames never get there, a bench or an
// optimisation loop does within its first frames.
#define DRT_JIT_AFTER_WORK ((uint64_t)1 << 31)

#define DRT_POLL_EVERY 4
#define DRT_TOTAL_WORDS 8           // segtotal: segments, queue rays read, written, capped paths, K2 rays, walked candidates

// a launch between its timing events (DRT_RENDER_TIMING), counted in the statistics
#define DRT_TIMED(S, K, ...)                                                       \
    do {                                                                           \
        int rc_t_;                                                                 \
        if ((rc_t_ = timing_begin((S).ctx, (S).timing, K)) != DRT_OK) return rc_t_; \
        __VA_ARGS__;                                                               \
        if ((rc_t_ = timing_end((S).ctx, (S).timing)) != DRT_OK) return rc_t_;     \
        (S).st->launches[K]++;                                                     \
    } while (0)

// ---- one shard's render: what the caller asked for, what the library decided, what both routes share -------------------
template <typename R>
struct Shard {
    typedef typename Q4<R>::T R4;
    typedef typename Q2<R>::T R2;
    // the request (render_launch)
    drt_hip_ctx* ctx;
    const drt_camera_desc* cam;
    const drt_render_params* rp;
    const float* d_adjoint;
    float* d_out_rgb;
    bool backward, timing;
    drt_hip_stats* st;
    uint32_t n_local_pixels;
    int D;                          // deepest vertex a path can reach
    double* film;
    int gimg_param;
    double* gfilm;
    float* d_out_gimg;
    // the scene in compute type R
    const DevScene<R>* d_scene;
    const R* d_params;
    DevBvh<R> bvh;
    // the plan (shard_plan)
    int spp;
    uint64_t total_paths;
    bool unbiased;                  // the reference's unbiased integration operator (integrate.hpp:39-52)
    bool loss_l2;                   // DRT_RENDER_LOSS_L2: every sample seeded with 2 (L_s - target_pixel), d_adjoint is the target
    hipFunction_t loss_kernel = nullptr;   // ... on k_path: the LOSS instantiation, made at run time only (drt_jit.h)
    bool can_fuse;                  // K2 folded into K3: analytic scenes, unless DRT_RENDER_UNFUSED asks for the textbook pipeline
    bool use_path, path_regen;      // the whole path in one launch (k_path); its regenerating form
    bool path_gen = false;          // ... its gradients in the general form: any number of parameters (DRT_NP_ANY)
    uint32_t gen_rows = 0, gen_clog2 = 0;
    bool mesh_path;                 // ... in a scene with a mesh: k_path_mesh (drt_path_mesh.h), the BVH walk inside the launch
    bool shade_tail;                // mesh scenes: the launch that produces a ray also intersects it with the analytic shapes and
                                    // builds the BVH walk's candidate lists (k_raygen / k_shade / k_adj_vertex <TAIL>)
    bool overlap_ok;                // this frame's k_path may run beside its neighbours' (its own stream and lane of partial sums)
    bool path_finish;               // one k_path launch covers the frame: ONE finishing launch writes image, gradients, totals
    uint32_t Pb, Sb;                // batch: pixels x samples
    size_t N;                       // batch capacity in paths
    uint32_t region_shift, region_size, max_regions;
    uint32_t path_spr, path_ranges;
    size_t path_waves;
    size_t cw;                      // count words of one batch
    DevBuf *fpart_buf, *gpart_buf, *counts_buf;   // the lane of k_path's partial sums this render uses
    ChainState<R> cs;               // unbiased operator: the chain vertices in HBM
    BatchArgs a;
    int n_fast, g_rows, g_stride;   // gradient partials: gpart[block][g_stride], rows [0, g_rows) reduced over the blocks by K7
    // typed views of the queue buffers
    int tail_nb = 1, tail_ring = 2; // shade_tail: stages of a shade launch (2: rays that miss the mesh bounds stay in registers through one more
                                    // vertex, DRT_HIP_TAIL_BOUNCES) and sets of queue / hit lanes that takes (3: a launch reads depth k, appends to k + 1, k + 2)
    R4* ra[3];                      // queue lanes: by depth parity, by depth mod tail_ring where the scene has a mesh (shade_tail)
    R2* rb[3];
    uint2* rid[3];
    HitRec<R>* hit;
    HitRec<R>* hitr[3];             // shade_tail: the hit lanes of depth d mod tail_ring
    int cset(int d) const { return tail_ring > 2 ? d & 1 : 0; }   // which set of candidate lists holds depth d (two sets only where launches have two stages)
    TailQueue<R> tailq(int d) const  // shade_tail: what a launch that appends to the queue of depth d gets
    {
        TailQueue<R> q;
        const int m = tail_ring;
        q.a = ra[d % m]; q.b = rb[d % m]; q.id = rid[d % m]; q.hit = hitr[d % m];
        const int c = cset(d);
        q.cand = (uint32_t*)ctx->cand[c].p; q.cand_a = (R4*)ctx->cand_a[c].p; q.cand_b = (R4*)ctx->cand_b[c].p;
        q.cand_count = (uint32_t*)ctx->cand_count[c].p;
        q.count = counts + (size_t)d * max_regions;
        return q;
    }
    R4* lacc;
    TapeRec<R>* tape;
    uint32_t* nv;
    uint32_t* counts;
    double *grad, *gpart;
    unsigned long long* totals;
    bool path_finished = false;

    // Bounces per fused launch (at most 8).  Inside a launch the lanes of ended paths idle -- cheap next to the queue traffic
    // saved, measured: even at absorb = 0.5 four bounces per launch beat one -- so a launch only stops where fewer than ~10 %
    // of its rays are expected to be left: ~7 % end per bounce on a miss or a light (Cornell-like scenes), the roulette removes
    // `absorb` of them at every depth >= min_bounces.  drt_render_params.bounces_per_launch (or DRT_HIP_SHADE_BOUNCES) forces n.
    int bounces_from(int k) const
    {
        if (!can_fuse)
            return 1;
        const int left = D - k;
        const int forced = tuning().shade_bounces > 0 ? tuning().shade_bounces : (rp->bounces_per_launch > 8 ? 8 : rp->bounces_per_launch);
        if (forced > 0)
            return forced < left ? forced : left;
        double alive = 1.0;
        int n = 0;
        while (n < left && n < 8) {
            alive *= 0.93 * ((k + n + 1) >= rp->min_bounces && (k + n + 1) < D ? 1.0 - rp->absorb : 1.0);
            ++n;
            if (alive < 0.1)
                break;
        }
        return n;
    }
};

// "k_path<float, SPEC, NP, NC, KindSig<...>, REGEN[, LOSS]>" / "k_path_unbiased<float, SPEC, NP, KindSig<...>>": the name expression
// of the f32 instantiation a render would launch, for the scene's own signature
inline std::string path_kernel_name(const drt_hip_ctx* ctx, bool tangents, bool unbiased, bool regen, bool loss, bool gen, bool f64 = false,
                                    bool gen_gimg = false)
{
    const std::string sg = drt_jit::sig_type(ctx->prog_sig, ctx->n_shapes);
    const char* sp = ctx->has_specular ? "true" : "false";
    const bool three = ctx->max_colour_param < 3;      // tangent state only for parameters that ARE some BxDF's colour
    char name[400];
    // (gen: the general form, any number of parameters: DRT_NP_ANY = -1)
    if (unbiased)
        snprintf(name, sizeof name, "k_path_unbiased<%s, %s, %d, %s>", f64 ? "double" : "float", sp, gen ? -1 : (ctx->n_params > 4 ? 8 : 4), sg.c_str());
    else {
        const int np = tangents ? (gen ? -1 : (ctx->n_params > 4 ? 8 : 4)) : 0;
        const int nc = tangents ? (gen ? (gen_gimg ? 1 : 0) : (ctx->n_params > 4 ? 8 : (three ? 3 : 4))) : 0;   // (general form: 1 = + the lanes' own sums of one row, the gradient image)
        snprintf(name, sizeof name, "k_path<%s, %s, %d, %d, %s, %s%s>", f64 ? "double" : "float", sp, np, nc, sg.c_str(), regen ? "true" : "false", loss ? ", true" : "");
    }
    return name;
}

// ---- which route, how large a batch, which grids ---------------------------------------------------------------------
template <typename R>
void shard_plan(Shard<R>& s)
{
    drt_hip_ctx* ctx = s.ctx;
    const drt_render_params* rp = s.rp;
    const int D = s.D;
    s.spp = rp->spp;
    s.total_paths = (uint64_t)s.n_local_pixels * (uint64_t)s.spp;
    s.unbiased = s.backward && (rp->flags & DRT_RENDER_UNBIASED) != 0 && s.gimg_param < 0;
    s.loss_l2 = s.backward && (rp->flags & DRT_RENDER_LOSS_L2) != 0;
    // K2 folded into K3 wherever nothing else consumes the hit records: never with a mesh (the BVH walk is its own kernel)
    s.can_fuse = !(rp->flags & DRT_RENDER_UNFUSED) && !ctx->has_mesh;
    // ---- k_path (drt_path.h): the whole path in one launch, in registers.  Taken when the scene is analytic and at most 8
    // parameters want gradients.  Two forms: lanes in lockstep (all at the same depth; a lane whose path ended idles to the end
    // of the sample) when most lanes stay busy to the end -- ~7 % of the paths end per bounce on a miss or a light, the
    // roulette removes `absorb` of the rest from min_bounces on -- and the regenerating form (a lane whose path ended starts
    // its next sample at once) otherwise: roulette-terminated paths under the default cap of 64, the reference's own
    // defaults (-b 1 -p 0.5).
    // Scenes with a mesh: k_path_mesh, the same launch with the BVH walk inside -- for SMALL frames (the frames of an
    // optimisation loop): one launch instead of ~20, 2-2.8 x the queue route's rate up to ~0.5 M camera samples, even at ~2 M,
    // half its rate at 16 M (a wave's lanes split between the vertex step and the walk: 20 of 64 per instruction;
    // profiles/r05_mesh_small_frames.txt, r05_mesh_fused_pmc.txt).  Biased operator, seeds linear in the radiance; the unbiased
    // operator, the per-sample loss and more than 8 parameters keep the queue wavefront.
    // (decided by the size of the FRAME, not of this shard's part of it: the shards of a frame take one route and tile it bit for bit)
    const bool mesh_ok = ctx->has_mesh && (long long)s.cam->width * s.cam->height * s.spp <= tuning().mesh_path_max && ctx->prog_sorted && !(rp->flags & DRT_RENDER_UNFUSED) && !s.unbiased &&
                         !s.loss_l2;
    // gradients: <= 8 parameters in registers / LDS columns; any number the kernels can stage (136: every analytic scene) through
    // the general form -- vertex history + per-wave tables (drt_path.h, DRT_NP_ANY); the gradient IMAGE (the lanes' own sums) in
    // analytic scenes too
    const bool grads_ok = !(s.backward || s.gimg_param >= 0) || ctx->n_params <= DRT_FAST_PARAMS ||
                          (ctx->n_params <= DRT_PATH_LDS_PARAMS && tuning().path_general && (s.gimg_param < 0 || !ctx->has_mesh));
    s.use_path = ((s.can_fuse && ctx->prog_ok) || mesh_ok) && D > 0 && grads_ok &&
                 rp->bounces_per_launch <= 0 && tuning().shade_bounces <= 0 && tuning().dump_path == -1;
    s.mesh_path = s.use_path && ctx->has_mesh;
    s.path_regen = tuning().path_regen > 0 || s.mesh_path;    // (k_path_mesh: every lane on its own, always)
    if (s.unbiased)
        s.path_regen = false;                  // (k_path_unbiased walks its samples in lockstep)
    const bool lane_is_pixel = s.gimg_param >= 0 && !s.mesh_path;   // the gradient image is the lanes' own sums: lockstep form only
    if (lane_is_pixel)
        s.path_regen = false;
    if (s.use_path && !s.mesh_path && !lane_is_pixel && tuning().path_regen < 0 && !s.unbiased) {
        // lockstep: a wave runs until the longest of its 64 paths ends -- the depth cap for fixed-depth renders, under the
        // roulette about the depth that 1 path in 256 reaches; regenerating: every lane runs the mean path length, at
        // ~1.7 x the cost per bounce (per-lane depth bookkeeping) + the camera code inside the loop.
        // Calibrated on the reference's scene at 512 x 512 x 64 (ms, lockstep / regenerating): depth 8 0.88 / 1.20,
        // 12: 1.26 / 1.81, 16: 1.62 / 2.33, 24: 2.35 / 3.29; -b 6 -p 0.1: 4.04 / 2.65, -b 2 -p 0.05: 5.57 / 3.37,
        // -b 3 -p 0.2: 2.41 / 1.47, -b 1 -p 0.5: 0.87 / 0.50; glossy, depth 16: 4.37 / 5.70.  In a closed room ~2.5 % of the
        // paths end per bounce on the light (measured mean lengths 7.3, 10.5, 13.3 at depths 8, 12, 16).
        double alive = 1.0, mean_len = 0.0;
        int longest = D;
        for (int k = 0; k < D; ++k) {
            mean_len += alive;
            alive *= 0.975 * ((k + 1) >= rp->min_bounces ? 1.0 - rp->absorb : 1.0);
            if (alive < 1.0 / 256 && longest == D)
                longest = k + 1;
        }
        s.path_regen = 1.7 * mean_len + 0.5 < (double)longest;
    }
    if (s.use_path && s.loss_l2) {
        // The per-sample seed 2 (L_s - target) needs the path's radiance before its gradients.  k_path has it where the path
        // ends on a light -- the only emissive vertex of a path unless some shape carries a BxDF AND an emitter -- in the LOSS
        // instantiation, which the library does not carry: it is compiled at run time (f32, unless the context may not
        // compile).  Everything else takes the tape route: two walks of the tape, k_radiance then k_backward.
        s.use_path = false;
        // (DRT_SPECIALISE_AUTO: no frame waits for the compiler -- the compile runs on the library's own thread from the first
        //  such frame on, and the tape route renders until it has delivered; DRT_SPECIALISE_NOW waits)
        if ((sizeof(R) == 4 || !ctx->user_header.empty()) && !ctx->emissive_bxdf && ctx->jit_mode >= DRT_SPECIALISE_AUTO) {
            const bool gen_loss = ctx->n_params > tuning().gen_above && ctx->n_params <= DRT_PATH_LDS_PARAMS && (tuning().path_general || ctx->n_params <= DRT_FAST_PARAMS);
            s.loss_kernel = jit_function(ctx, path_kernel_name(ctx, true, false, s.path_regen, true, gen_loss, sizeof(R) == 8),
                                         ctx->jit_mode > DRT_SPECIALISE_AUTO || !ctx->user_header.empty());
            s.use_path = s.loss_kernel != nullptr;
        }
    }
    // (the general form also where the register form would do but is slower: its 5 ... 8-parameter instantiation keeps 24 LDS
    //  columns per thread -- 0.78 ms against the general form's 0.75 on config 3's frame with an albedo per wall)
    s.path_gen = s.use_path && (s.backward || s.gimg_param >= 0) && ctx->n_params <= DRT_PATH_LDS_PARAMS &&
                 (s.gimg_param >= 0 ? ctx->n_params > DRT_FAST_PARAMS && !s.mesh_path       // (the image keeps the column form where it exists)
                                    : ctx->n_params > tuning().gen_above && (tuning().path_general || ctx->n_params <= DRT_FAST_PARAMS));
    // Batch = the paths that are in flight at once on the queue route.  The BVH walk wants it LARGE: its launches end in a
    // tail of ~0.1 ms whatever their size (the list counters run dry, every wave finishes what it holds), so config 4 at full
    // size (1024^2 x 256 spp) takes 115 / 101 / 98 / 96 ms with 2^24 / 2^26 / 2^27 / 2^28 paths per batch and one GPU's
    // share of it (33.5 M paths) 14.3 ms in two batches, 13.1 in one.  Every path in flight owns ~0.2 KB of queue lanes,
    // tape and candidate records (twice that in f64): the default is the largest power of two whose buffers fit in an eighth
    // of the device's memory, at most 32 GB -- 2^27 paths (27 GB) for a depth-8 f32 render on a 288 GB part.
    uint64_t cap_default;
    {
        const uint64_t f = sizeof(R) / 4;
        // (mesh scenes: one set of candidate records, 36 B; the unbiased operator's two-stage shade launches: a third set of queue
        //  and hit lanes and a second set of candidate records, + 76 B)
        const uint64_t per_path = f * (112u + 8u * (uint64_t)(D > 0 ? D : 1) + (ctx->has_mesh ? 36u : 0u) + (s.unbiased ? (ctx->has_mesh ? 186u : 110u) : 0u)) + 24u;
        const uint64_t budget = std::min<uint64_t>(ctx->device_mem / 8, (uint64_t)32 << 30);
        cap_default = (uint64_t)1 << 22;
        while (cap_default < ((uint64_t)1 << 28) && 2 * cap_default * per_path <= budget)
            cap_default *= 2;
        // (a device that other work has filled: no more than half of what is free now, unless the buffers exist already)
        if (!s.use_path && rp->batch_paths <= 0 && (uint64_t)ctx->ray_a[0].cap < std::min<uint64_t>(cap_default, s.total_paths) * 16u * f) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                while (cap_default > ((uint64_t)1 << 22) && cap_default * per_path > (uint64_t)free_b / 2)
                    cap_default /= 2;
        }
    }
    uint64_t cap = rp->batch_paths > 0 ? (uint64_t)rp->batch_paths : cap_default;
    if (s.use_path && rp->batch_paths <= 0)
        cap = s.total_paths;               // no per-path memory: one batch covers the frame
    if (tuning().batch_paths > 0)
        cap = (uint64_t)tuning().batch_paths;
    if (cap > s.total_paths) cap = s.total_paths;
    if (cap < 1) cap = 1;
    if (cap > 0x7FFFFFFFull) cap = 0x7FFFFFFFull;
    s.Pb = (uint32_t)(cap / (uint64_t)s.spp);
    if (s.Pb < 1) s.Pb = 1;
    if (s.Pb > s.n_local_pixels) s.Pb = s.n_local_pixels;
    s.Sb = (uint32_t)(cap / s.Pb);
    if (s.Sb > (uint32_t)s.spp) s.Sb = (uint32_t)s.spp;
    if (s.Sb < 1) s.Sb = 1;
    s.N = (size_t)s.Pb * s.Sb;
    // queue regions: one wave each; enough of them to fill 256 CUs several times over
    s.region_shift = 8;          // 256 slots: 4 chunks per wave (sweep in profiles/: 64..4096)
    if (tuning().region_size > 0)
        for (s.region_shift = 6; s.region_shift < 20 && (1l << s.region_shift) < (long)tuning().region_size; ++s.region_shift) { }
    while (s.region_shift > 6 && (s.N >> s.region_shift) < (size_t)ctx->n_cu * 32)
        --s.region_shift;
    s.region_size = 1u << s.region_shift;
    s.max_regions = (uint32_t)((s.N + s.region_size - 1) / s.region_size);
    // k_path geometry: wave <-> (64 pixels, spr samples); enough waves for ~5-6 rounds of what the chip holds (the tail stays
    // short) in ranges of equal length (sweep on config 3, ms per launch: 16 samples per range 0.827, 13: 0.818, 10: 0.793,
    // 8: 0.812, 7: 0.795, 4: 0.811); regenerating lanes balance themselves over their sample range: longer ranges, fewer waves
    const uint32_t path_groups = (s.Pb + DRT_WAVE - 1) / DRT_WAVE;
    {
        const uint64_t target = (uint64_t)ctx->n_cu * (s.path_regen ? 32 : 112);
        const uint64_t want = std::max<uint64_t>(1, (target + path_groups - 1) / path_groups);   // ranges
        s.path_spr = (uint32_t)((s.Sb + want - 1) / want);
        if (tuning().path_spr > 0)
            s.path_spr = (uint32_t)tuning().path_spr;
        if (s.path_spr < 1) s.path_spr = 1;
        if (s.path_spr > s.Sb) s.path_spr = s.Sb;
    }
    s.path_ranges = (s.Sb + s.path_spr - 1) / s.path_spr;
    s.path_waves = (size_t)path_groups * s.path_ranges;
    s.shade_tail = ctx->has_mesh;
    // two-stage shade launches (measured, same process, alternating): config 4's share 5.90 -> 6.15 ms of shade launches (20 % fewer
    // bytes, 18 % more instructions, the second stage at two lanes in three), the unbiased operator's rounds 14.05 -> 13.25: on for those
    // (this knob alone is read at every render: the two settings differ by a few per cent, less than one process differs from the
    //  next on a box that warms up -- tools/tail_check.py alternates them inside one process)
    const int tail_bounces = tail_bounces_now();
    s.tail_nb = s.shade_tail && (tail_bounces > 1 || (tail_bounces == 0 && s.unbiased)) ? 2 : 1;
    s.tail_ring = s.tail_nb > 1 ? 3 : 2;
    s.overlap_ok = ctx->overlap_next && s.use_path && !s.timing && s.gimg_param < 0 && ctx->path_stream[0] && ctx->ev_copied[0];
    const bool odd = s.overlap_ok && (ctx->slot & 1);
    s.fpart_buf = odd ? &ctx->fpart2 : &ctx->fpart;
    s.gpart_buf = odd ? &ctx->gpart2 : &ctx->gpart;
    s.counts_buf = odd ? &ctx->counts2 : &ctx->counts;
    s.cw = s.use_path ? (s.mesh_path ? 3 : 2) * s.path_waves      // [segments | capped paths (| rays the BVH walk took)] per wave
                      : (size_t)(D + 2) * s.max_regions;          // counts[depth][region] of one batch (row D: capped paths, D + 1: rays kept in registers)
    // a k_path launch that covers the whole frame is followed by ONE finishing launch that WRITES image, gradients and
    // totals (k_path_finish); every other route accumulates into zeroed buffers
    s.path_finish = s.use_path && s.Pb == s.n_local_pixels && s.Sb == (uint32_t)s.spp && (!s.film || s.d_out_rgb);
    if (s.path_gen) {
        // a wave's table holds DRT_GEN_TABLE elements: as many copies of every row as fit, at most 16 (same-address LDS atomics
        // of one instruction serialise; with 16 copies the ~12 lanes of a wave that end a sample on a light rarely meet)
        s.gen_rows = (uint32_t)std::max(1, ctx->n_grad_slots) * 3u;
        s.gen_clog2 = 0;
        while (s.gen_clog2 < 4 && (s.gen_rows << (s.gen_clog2 + 1)) <= DRT_GEN_TABLE)
            ++s.gen_clog2;
        if (tuning().gen_copies_log2 >= 0 && (s.gen_rows << tuning().gen_copies_log2) <= DRT_GEN_TABLE)
            s.gen_clog2 = (uint32_t)tuning().gen_copies_log2;
    }
    s.n_fast = ctx->n_params < DRT_FAST_PARAMS ? ctx->n_params : DRT_FAST_PARAMS;
    const bool g_general = ctx->n_params > DRT_FAST_PARAMS;
    s.g_rows = g_general ? std::min(ctx->n_params, DRT_LDS_PARAMS) * 3 : s.n_fast * 3;
    s.g_stride = g_general ? s.g_rows : DRT_FAST_PARAMS * 3;
}

// ---- the device buffers of the plan ------------------------------------------------------------------------------------
template <typename R>
int shard_buffers(Shard<R>& s)
{
    typedef typename Shard<R>::R4 R4;
    typedef typename Shard<R>::R2 R2;
    drt_hip_ctx* ctx = s.ctx;
    const size_t N = s.N;
    const int D = s.D;
    int rc;
    memset(&s.cs, 0, sizeof s.cs);
    if (s.use_path) {
        if ((rc = ensure(ctx, *s.fpart_buf, (size_t)s.path_ranges * 3 * s.Pb * sizeof(double))) != DRT_OK) return rc;
        if (s.gimg_param >= 0)
            if ((rc = ensure(ctx, ctx->gpix, (size_t)s.path_ranges * 3 * s.Pb * sizeof(double))) != DRT_OK) return rc;
        if (s.mesh_path) {   // the traversal stack's entries beyond the ones in LDS, per thread of the grid (one area per k_path stream)
            const size_t threads = ((s.path_waves + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE)) * DRT_BLOCK;
            if ((rc = ensure(ctx, ctx->mesh_ovf[(s.overlap_ok && (ctx->slot & 1)) ? 1 : 0],
                             threads * (size_t)(DRT_BVH_STACK - DRT_MESH_LDS_STACK) * sizeof(uint32_t))) != DRT_OK) return rc;
        }
    } else {
        for (int i = 0; i < (s.shade_tail ? s.tail_ring : 2); ++i) {
            if ((rc = ensure(ctx, ctx->ray_a[i], N * sizeof(R4))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ray_b[i], N * sizeof(R2))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ray_id[i], N * sizeof(uint2))) != DRT_OK) return rc;
        }
        if ((rc = ensure(ctx, ctx->hit, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
        if (ctx->has_mesh) {
            // the hit lane double-buffered like the queue; the rays the BVH walk has to see: one dense list of complete records
            // per queue region (the region's own span of the candidate arrays) + the walk's list counters
            const size_t cand_words = (size_t)s.max_regions * s.region_size;
            if ((rc = ensure(ctx, ctx->hit2, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
            if (s.tail_ring > 2)
                if ((rc = ensure(ctx, ctx->hit3, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
            for (int i = 0; i < (s.tail_ring > 2 ? 2 : 1); ++i) {
                if ((rc = ensure(ctx, ctx->cand[i], cand_words * sizeof(uint32_t))) != DRT_OK) return rc;
                if ((rc = ensure(ctx, ctx->cand_a[i], cand_words * sizeof(R4))) != DRT_OK) return rc;
                if ((rc = ensure(ctx, ctx->cand_b[i], cand_words * sizeof(R4))) != DRT_OK) return rc;
                if ((rc = ensure(ctx, ctx->cand_count[i], ((size_t)s.max_regions + DRT_PULL_WORDS) * sizeof(uint32_t))) != DRT_OK) return rc;
            }
        }
        if ((rc = ensure(ctx, ctx->lacc, N * sizeof(R4))) != DRT_OK) return rc;
        if (s.gimg_param >= 0)
            if ((rc = ensure(ctx, ctx->gpath, N * sizeof(R4))) != DRT_OK) return rc;
        if (s.unbiased) {
            if ((rc = ensure(ctx, ctx->ch_cva, N * sizeof(R4))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_cvb, N * sizeof(R2))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_cvh, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_nxa, N * sizeof(R4))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_nxb, N * sizeof(R2))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_nxh, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_g, N * sizeof(R4))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_w, N * sizeof(R4))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_ids, N * sizeof(uint32_t))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_ndraw, N * sizeof(uint32_t))) != DRT_OK) return rc;
            if ((rc = ensure(ctx, ctx->ch_dbase, N * sizeof(uint32_t))) != DRT_OK) return rc;
            s.cs.cv_a = (R4*)ctx->ch_cva.p; s.cs.cv_b = (R2*)ctx->ch_cvb.p; s.cs.cv_hit = (HitRec<R>*)ctx->ch_cvh.p;
            s.cs.nx_a = (R4*)ctx->ch_nxa.p; s.cs.nx_b = (R2*)ctx->ch_nxb.p; s.cs.nx_hit = (HitRec<R>*)ctx->ch_nxh.p;
            s.cs.g = (R4*)ctx->ch_g.p; s.cs.w = (R4*)ctx->ch_w.p;
            s.cs.ids = (uint32_t*)ctx->ch_ids.p; s.cs.ndraw = (uint32_t*)ctx->ch_ndraw.p; s.cs.dbase = (uint32_t*)ctx->ch_dbase.p;
        }
        if ((rc = ensure(ctx, ctx->tape, N * sizeof(TapeRec<R>) * (size_t)(D > 0 ? D : 1))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->nv, N * sizeof(uint32_t))) != DRT_OK) return rc;
    }
    if ((rc = ensure(ctx, *s.counts_buf, s.cw * sizeof(uint32_t))) != DRT_OK) return rc;
    if ((rc = ensure(ctx, ctx->segtotal[ctx->slot], DRT_TOTAL_WORDS * sizeof(unsigned long long))) != DRT_OK) return rc;
    if (s.backward) {   // per-block partial sums: K6's persistent grid, the shade kernel's one block per 4 regions, or k_path's blocks
        const size_t shade_blocks = (s.max_regions + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE);
        size_t blocks = std::max<size_t>(shade_blocks, (size_t)grid_for(ctx, N));
        const size_t path_blocks = (s.path_waves + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE);
        if (s.use_path && path_blocks > blocks)
            blocks = path_blocks;
        // rows per block: 24 for the register paths (<= 8 parameters), else one per parameter channel (LDS accumulators)
        const size_t rows = s.path_gen ? (size_t)s.gen_rows
                                       : (ctx->n_params <= DRT_FAST_PARAMS ? (size_t)DRT_FAST_PARAMS * 3
                                                                           : (size_t)std::min(ctx->n_params, DRT_LDS_PARAMS) * 3);
        if ((rc = ensure(ctx, *s.gpart_buf, blocks * rows * sizeof(double))) != DRT_OK) return rc;
    }
    for (int i = 0; i < 3; ++i) {
        s.ra[i] = (R4*)ctx->ray_a[i].p;
        s.rb[i] = (R2*)ctx->ray_b[i].p;
        s.rid[i] = (uint2*)ctx->ray_id[i].p;
    }
    s.hit = (HitRec<R>*)ctx->hit.p;
    s.hitr[0] = s.hit; s.hitr[1] = (HitRec<R>*)ctx->hit2.p; s.hitr[2] = (HitRec<R>*)ctx->hit3.p;
    s.lacc = (R4*)ctx->lacc.p;
    s.tape = (TapeRec<R>*)ctx->tape.p;
    s.nv = (uint32_t*)ctx->nv.p;
    s.counts = (uint32_t*)s.counts_buf->p;          // reused by every batch (stream order)
    s.grad = (double*)ctx->grad[ctx->slot].p;
    s.gpart = (double*)s.gpart_buf->p;
    s.totals = (unsigned long long*)ctx->segtotal[ctx->slot].p;
    return DRT_OK;
}

// ---- the constants of the frame as the kernels take them -----------------------------------------------------------------
template <typename R>
void shard_args(Shard<R>& s)
{
    BatchArgs& a = s.a;
    const drt_camera_desc* cam = s.cam;
    const drt_render_params* rp = s.rp;
    memset(&a, 0, sizeof a);
    a.W = cam->width; a.H = cam->height; a.spp = s.spp;
    a.shard = rp->n_shards > 1 ? rp->shard : 0;
    a.n_shards = rp->n_shards > 1 ? rp->n_shards : 1;
    a.band = rp->band_rows > 0 ? rp->band_rows : 1;
    a.min_bounces = rp->min_bounces;
    a.depth_cap = s.D;
    a.cap_is_roulette = (rp->absorb >= 1.0 && rp->min_bounces == s.D) ? 1 : 0;
    a.cap_draws = (a.cap_is_roulette || rp->max_depth <= 0) ? 1 : 0;
    a.absorb = rp->absorb;
    a.seed = rp->seed;
    a.rng_stream = drt_rng_stream(rp->seed, 0u);
    for (int i = 0; i < 3; ++i) {
        a.eye[i] = cam->eye[i]; a.fwd[i] = cam->forward[i];
        a.right[i] = cam->right[i]; a.up[i] = cam->up[i];
    }
    a.region_size = s.region_size;
    a.region_shift = s.region_shift;
    a.bvh_refill = tuning().bvh_refill >= 0 ? (uint32_t)tuning().bvh_refill : DRT_BVH_REFILL;
    a.bvh_descend_min = tuning().bvh_descend_min >= 0 ? (uint32_t)tuning().bvh_descend_min : DRT_BVH_DESCEND_MIN;
    {   // smallest r with !(double(r) / RAND_MAX < absorb): the roulette test as an integer compare
        double guess = floor(rp->absorb * DRT_RAND_MAX_D);
        int64_t r = (int64_t)guess - 2;
        if (r < 0) r = 0;
        while (r <= 2147483647LL && (double)r / DRT_RAND_MAX_D < rp->absorb)
            ++r;
        a.rr_threshold = (uint32_t)r;
    }
    a.tan_half = tan(cam->vfov / 2.);
    a.aspect = (double)cam->width / (double)cam->height;
}

// ---- route 1: the whole batch in ONE launch: camera -> path -> radiance sums + gradient partials (k_path) ----------------
template <typename R>
int path_batch(Shard<R>& s)
{
    drt_hip_ctx* ctx = s.ctx;
    const drt_render_params* rp = s.rp;
    const BatchArgs& a = s.a;
    drt_hip_stats* st = s.st;
    const bool backward = s.backward, unbiased = s.unbiased, path_regen = s.path_regen;
    PathArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.W = a.W; pa.H = a.H; pa.spp = a.spp;
    pa.shard = a.shard; pa.n_shards = a.n_shards; pa.band = a.band;
    pa.Pb = a.Pb; pa.p0 = a.p0; pa.Sb = a.Sb; pa.s0 = a.s0;
    pa.spr = s.path_spr < a.Sb ? s.path_spr : a.Sb;
    pa.n_ranges = (a.Sb + pa.spr - 1) / pa.spr;
    pa.n_groups = (a.Pb + DRT_WAVE - 1) / DRT_WAVE;
    pa.min_bounces = a.min_bounces; pa.depth_cap = a.depth_cap; pa.cap_is_roulette = a.cap_is_roulette; pa.cap_draws = a.cap_draws;
    pa.rr_threshold = a.rr_threshold; pa.seed = a.seed; pa.rng_stream = a.rng_stream;
    pa.regen_min = (uint32_t)tuning().path_regen_min;
    pa.shade_min = (uint32_t)tuning().mesh_shade_min;
    pa.descend_min = a.bvh_descend_min;
    pa.p_rr = 1.0 - rp->absorb;
    pa.inv_p_rr = rp->absorb < 1.0 ? 1.0 / (1.0 - rp->absorb) : 0.0;   // (never used when every path ends at min_bounces)
    pa.p_rr_f = (float)pa.p_rr; pa.inv_p_rr_f = (float)pa.inv_p_rr;
    for (int i = 0; i < 3; ++i) {
        pa.eye[i] = a.eye[i]; pa.fwd[i] = a.fwd[i]; pa.right[i] = a.right[i]; pa.up[i] = a.up[i];
    }
    pa.tan_half = a.tan_half; pa.aspect = a.aspect;
    pa.inv_W = 1.0 / (double)a.W; pa.inv_H = 1.0 / (double)a.H;
    for (int i = 0; i < 3; ++i) {
        pa.eye_f[i] = (float)pa.eye[i]; pa.fwd_f[i] = (float)pa.fwd[i]; pa.right_f[i] = (float)pa.right[i]; pa.up_f[i] = (float)pa.up[i];
    }
    pa.cs_step_f = (float)(2. * pa.aspect * pa.tan_half * pa.inv_W);
    pa.ct_step_f = (float)(2. * pa.tan_half * pa.inv_H);
    pa.gimg_param = s.gimg_param;
    pa.gen_rows = s.gen_rows; pa.gen_clog2 = s.gen_clog2;
    // the general form's vertex history: a word per four vertices and thread, in dynamic shared memory
    const bool gen = s.path_gen;
    // (the first words in LDS, as many as leave the kernel's blocks per CU alone: four in the lockstep k_path -- 16 vertices --,
    //  none in the regenerating forms, whose static LDS sits right under a block's share; the others in global memory)
    const uint32_t hist_words = gen ? (uint32_t)(a.depth_cap / 4) : 0u;
    pa.hist_lds = std::min<uint32_t>(hist_words, (path_regen || s.mesh_path) ? 0u : 4u);
    if (tuning().gen_hist_lds >= 0)
        pa.hist_lds = std::min<uint32_t>(hist_words, (uint32_t)tuning().gen_hist_lds);
    const unsigned hist_bytes = pa.hist_lds * DRT_BLOCK * (unsigned)sizeof(uint32_t);
    const uint32_t hist_ovf_words = hist_words - pa.hist_lds;
    const unsigned short* slot_map = gen ? (const unsigned short*)((const char*)s.d_scene + offsetof(DevScene<R>, grad_slot)) : (const unsigned short*)nullptr;
    const int g_rows = gen ? (int)s.gen_rows : s.n_fast * 3, g_stride = gen ? (int)s.gen_rows : DRT_FAST_PARAMS * 3;
    const size_t n_waves = (size_t)pa.n_groups * pa.n_ranges;
    const int gpath = (int)((n_waves + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE));
    const DevScene<R>* d_scene = s.d_scene;
    const R* d_params = s.d_params;
    const float* d_adjoint = s.d_adjoint;
    double* gpart = s.gpart;
    uint32_t* counts = s.counts;
    double* fpart = s.film ? (double*)s.fpart_buf->p : (double*)nullptr;
    double* gpix = s.gimg_param >= 0 ? (double*)ctx->gpix.p : (double*)nullptr;   // gradient image partials
    // The closest-hit program.  The kinds of the reference's own scene are compiled in, in the instantiation the library
    // carries (f64 too: the verification mode runs the same program with full-precision reciprocals and square roots); any
    // other analytic scene reads its kinds at run time (the kind-sorted program) until it has rendered enough for a kernel
    // of its own to pay (drt_jit.h; f32 only: the f64 mode keeps the reference's literal shape loop for such scenes).
    const bool builtin = tuning().builtin_program && ctx->jit_mode >= 0 && ctx->n_shapes == DRT_NSIG_CORNELL &&
                         ctx->prog_sig[0] == DRT_SIG_CORNELL && !s.loss_kernel && !s.mesh_path &&
                         ctx->user_header.empty();      // (caller-defined kinds -- a BxDF on the reference's own shapes -- exist in hiprtc's kernel only)
    unsigned long long* ptotal = s.path_finish ? s.totals : (unsigned long long*)nullptr;
    // (frames that overlap: this frame's grid goes to the lane's own stream, behind whoever still uses the lane's buffers, and
    //  the finishing launch on the context's stream waits for it.  Scene uploads block until they are done; a parameter update
    //  is a launch in the context's stream that nobody waits for on the host -- the first frame of either lane behind it waits
    //  for its event --; an adjoint image the caller may have produced in the context's stream's order: then the frame keeps
    //  its place in it.)
    hipStream_t ks = ctx->stream;
    const bool overlap = s.overlap_ok && s.path_finish;
    const int lane2 = ctx->slot & 1;                // which of the two k_path streams / sets of partial sums
    if (overlap) {
        ks = ctx->path_stream[lane2];
        if (d_adjoint) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_begin[lane2], ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ks, ctx->ev_begin[lane2], 0));
        }
        // (the lane's buffers: their last user -- this lane's previous frame, or a render that went through the context's
        //  stream -- has enqueued its last reader on the context's stream by the time its event is recorded)
        if (ctx->lane_used[lane2] && ctx->ev_lane_free[lane2])
            HIPCHK(ctx, hipStreamWaitEvent(ks, ctx->ev_lane_free[lane2], 0));
        if (ctx->params_pending[lane2] && ctx->ev_params) {
            HIPCHK(ctx, hipStreamWaitEvent(ks, ctx->ev_params, 0));
            ctx->params_pending[lane2] = false;
        }
    }
    const bool three = ctx->max_colour_param < 3;
    const bool tangents = backward || s.gimg_param >= 0;
    hipFunction_t jit = s.loss_kernel;
    ctx->scene_work += (uint64_t)a.n_paths * (uint64_t)(s.D > 0 ? s.D : 1);
    if (!jit && !builtin && !s.mesh_path && ctx->jit_mode > 0 && sizeof(R) == 4 && ctx->user_header.empty() &&
        (ctx->jit_mode > 1 || ctx->scene_work >= DRT_JIT_AFTER_WORK))
        jit = jit_function(ctx, path_kernel_name(ctx, tangents, unbiased, path_regen, false, s.path_gen, false, s.path_gen && s.gimg_param >= 0), ctx->jit_mode > 1);
    if (!ctx->user_header.empty() && !jit) {
        // caller-defined shape kinds: their code exists only in a kernel compiled for this scene -- made now, waited for, in either
        // compute type (shard_plan has checked that the context may compile)
        jit = jit_function(ctx, path_kernel_name(ctx, tangents, unbiased, path_regen, false, s.path_gen, sizeof(R) == 8, s.path_gen && s.gimg_param >= 0), true);
        if (!jit)
            return fail(ctx, DRT_ERR_UNSUPPORTED, ("render: the scene's caller-defined shape kinds did not compile: " + ctx->jit_error).c_str());
    }
    st->path_program = builtin ? DRT_PROGRAM_BUILTIN : (jit ? DRT_PROGRAM_SPECIALISED : DRT_PROGRAM_SORTED);
    int rc;
    if ((rc = timing_begin(ctx, s.timing, DRT_K_PATH)) != DRT_OK) return rc;
#define DRT_LAUNCH_PATH(SPEC, NP, NC, SG)                                                                                 \
    do {                                                                                                                 \
        if (path_regen)                                                                                                  \
            hipLaunchKernelGGL((k_path<R, SPEC, NP, NC, SG, true>), dim3(gpath), dim3(DRT_BLOCK), hist_bytes, ks,        \
                               pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal, gpix);                   \
        else                                                                                                             \
            hipLaunchKernelGGL((k_path<R, SPEC, NP, NC, SG, false>), dim3(gpath), dim3(DRT_BLOCK), hist_bytes, ks,       \
                               pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal, gpix);                   \
    } while (0)
#define DRT_LAUNCH_PATH_SIG(SPEC, NP, NC)                                                  \
    do {                                                                                   \
        if (builtin) DRT_LAUNCH_PATH(SPEC, NP, NC, SigCornell);                            \
        else DRT_LAUNCH_PATH(SPEC, NP, NC, SigNone);                                       \
    } while (0)
#define DRT_LAUNCH_UNB(SPEC, NP)                                                                                              \
    do {                                                                                                                      \
        if (builtin)                                                                                                          \
            hipLaunchKernelGGL((k_path_unbiased<R, SPEC, NP, SigCornell>), dim3(gpath), dim3(DRT_BLOCK), 0,                    \
                               ks, pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal);                            \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_path_unbiased<R, SPEC, NP, SigNone>), dim3(gpath), dim3(DRT_BLOCK), 0, ks, pa,               \
                               d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal);                                    \
    } while (0)
    uint32_t* ovf = s.mesh_path ? (uint32_t*)ctx->mesh_ovf[(s.overlap_ok && lane2) ? 1 : 0].p : (uint32_t*)nullptr;
    const uint32_t ovf_stride = (uint32_t)gpath * DRT_BLOCK;
    pa.hist_stride = ovf_stride;
    if (gen && hist_ovf_words > 0) {
        // the general form's history beyond its LDS words: a column per thread of the grid (one area per k_path stream); the
        // mesh kernel keeps it behind its traversal stacks' global part
        DevBuf& hb = s.mesh_path ? ctx->mesh_ovf[(s.overlap_ok && lane2) ? 1 : 0] : ctx->hist_ovf[(s.overlap_ok && lane2) ? 1 : 0];
        const size_t words = (size_t)hist_ovf_words + (s.mesh_path ? (size_t)(DRT_BVH_STACK - DRT_MESH_LDS_STACK) : 0);
        int rce;
        if ((rce = ensure(ctx, hb, words * ovf_stride * sizeof(uint32_t))) != DRT_OK) return rce;
        if (s.mesh_path) {
            ovf = (uint32_t*)hb.p;
            pa.hist_ovf = (unsigned long long)(uintptr_t)(ovf + (size_t)(DRT_BVH_STACK - DRT_MESH_LDS_STACK) * ovf_stride);
        } else
            pa.hist_ovf = (unsigned long long)(uintptr_t)hb.p;
    }
    // the gradient image in the general form: the row of its parameter (none: the parameter requires no gradient -- its image is zero)
    pa.gimg_row = DRT_SLOT_NONE;
    if (gen && s.gimg_param >= 0 && s.gimg_param < DRT_PATH_LDS_PARAMS && ctx->requires_grad[(size_t)s.gimg_param]) {
        uint32_t row = 0;
        for (int p2 = 0; p2 < s.gimg_param; ++p2)
            row += ctx->requires_grad[(size_t)p2] ? 1u : 0u;
        pa.gimg_row = row;
    }
    const DevBvh<R> bvh = s.bvh;
#define DRT_LAUNCH_MESH(SPEC, NP, NC)                                                                              \
    hipLaunchKernelGGL((k_path_mesh<R, SPEC, NP, NC>), dim3(gpath), dim3(DRT_BLOCK), hist_bytes, ks, pa, d_scene, d_params, \
                       d_adjoint, bvh, ovf, ovf_stride, gpart, fpart, counts, ptotal, gpix)
    if (s.mesh_path) {
        if (tangents && gen) { if (ctx->has_specular) DRT_LAUNCH_MESH(true, DRT_NP_ANY, 0); else DRT_LAUNCH_MESH(false, DRT_NP_ANY, 0); }
        else if (tangents && ctx->n_params > 4) { if (ctx->has_specular) DRT_LAUNCH_MESH(true, 8, 8); else DRT_LAUNCH_MESH(false, 8, 8); }
        else if (tangents && three) { if (ctx->has_specular) DRT_LAUNCH_MESH(true, 4, 3); else DRT_LAUNCH_MESH(false, 4, 3); }
        else if (tangents) { if (ctx->has_specular) DRT_LAUNCH_MESH(true, 4, 4); else DRT_LAUNCH_MESH(false, 4, 4); }
        else { if (ctx->has_specular) DRT_LAUNCH_MESH(true, 0, 0); else DRT_LAUNCH_MESH(false, 0, 0); }
    } else
#undef DRT_LAUNCH_MESH
    if (jit) {
        void* args_path[] = {&pa, &d_scene, &d_params, &d_adjoint, &gpart, &fpart, &counts, &ptotal, &gpix};
        void* args_unb[] = {&pa, &d_scene, &d_params, &d_adjoint, &gpart, &fpart, &counts, &ptotal};
        HIPCHK(ctx, hipModuleLaunchKernel(jit, (unsigned)gpath, 1, 1, DRT_BLOCK, 1, 1, unbiased ? 0u : hist_bytes, ks, unbiased ? args_unb : args_path, nullptr));
    } else if (unbiased) {                      // the unbiased operator: fresh suffix paths per vertex, in registers
        if (gen) { if (ctx->has_specular) DRT_LAUNCH_UNB(true, DRT_NP_ANY); else DRT_LAUNCH_UNB(false, DRT_NP_ANY); }
        else if (ctx->n_params > 4) { if (ctx->has_specular) DRT_LAUNCH_UNB(true, 8); else DRT_LAUNCH_UNB(false, 8); }
        else { if (ctx->has_specular) DRT_LAUNCH_UNB(true, 4); else DRT_LAUNCH_UNB(false, 4); }
    } else if (tangents && gen && s.gimg_param >= 0) {   // ... and the gradient image of one of them (lockstep: a lane is a pixel)
#define DRT_LAUNCH_GIMG(SPEC, SG) hipLaunchKernelGGL((k_path<R, SPEC, DRT_NP_ANY, 1, SG, false>), dim3(gpath), dim3(DRT_BLOCK), hist_bytes, ks, \
                                                     pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal, gpix)
        if (ctx->has_specular) { if (builtin) DRT_LAUNCH_GIMG(true, SigCornell); else DRT_LAUNCH_GIMG(true, SigNone); }
        else { if (builtin) DRT_LAUNCH_GIMG(false, SigCornell); else DRT_LAUNCH_GIMG(false, SigNone); }
#undef DRT_LAUNCH_GIMG
    } else if (tangents && gen) {                      // any number of parameters
        if (ctx->has_specular) DRT_LAUNCH_PATH_SIG(true, DRT_NP_ANY, 0);
        else DRT_LAUNCH_PATH_SIG(false, DRT_NP_ANY, 0);
    } else if (tangents && ctx->n_params > 4) {        // 5 .. 8 parameters
        if (ctx->has_specular) DRT_LAUNCH_PATH_SIG(true, 8, 8);
        else DRT_LAUNCH_PATH_SIG(false, 8, 8);
    } else if (tangents) {
        if (ctx->has_specular) { if (three) DRT_LAUNCH_PATH_SIG(true, 4, 3); else DRT_LAUNCH_PATH_SIG(true, 4, 4); }
        else { if (three) DRT_LAUNCH_PATH_SIG(false, 4, 3); else DRT_LAUNCH_PATH_SIG(false, 4, 4); }
    } else {
        if (ctx->has_specular) DRT_LAUNCH_PATH_SIG(true, 0, 0);
        else DRT_LAUNCH_PATH_SIG(false, 0, 0);
    }
#undef DRT_LAUNCH_UNB
#undef DRT_LAUNCH_PATH_SIG
#undef DRT_LAUNCH_PATH
    if ((rc = timing_end(ctx, s.timing)) != DRT_OK) return rc;
    if (overlap) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_path[lane2], ks));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_path[lane2], 0));
    }
    st->launches[DRT_K_PATH]++;
    st->path_bytes += (s.film ? (uint64_t)pa.n_ranges * a.Pb * 3 * sizeof(double) : 0) +
                      (backward ? (uint64_t)gpath * (uint64_t)g_stride * sizeof(double) : 0) + (s.mesh_path ? 3 : 2) * n_waves * sizeof(uint32_t);
    if (s.path_finish) {
        // image, gradients and totals of the frame in one launch (timed in the film slot)
        const uint32_t film_blocks = s.film ? (uint32_t)grid_for(ctx, a.Pb) : 0u;
        const uint32_t grad_words = backward ? (uint32_t)ctx->n_params * 3u : 0u;
        const uint32_t count_blocks = (uint32_t)std::min<size_t>(64, (n_waves + DRT_BLOCK - 1) / DRT_BLOCK);
        DRT_TIMED(s, DRT_K_FILM,
                  hipLaunchKernelGGL(k_path_finish, dim3(film_blocks + grad_words + count_blocks), dim3(DRT_BLOCK), 0, ctx->stream, pa,
                                     (const double*)fpart, s.d_out_rgb, film_blocks, (const double*)gpart, gpath, g_rows,
                                     g_stride, s.grad, grad_words, (const uint32_t*)counts, (uint32_t)n_waves, s.totals,
                                     s.mesh_path ? 3u : 2u, slot_map));
        st->units[DRT_K_FILM] += a.n_paths;
        if (gpix && s.d_out_gimg) {   // the gradient image: the same sums over the sample ranges, its own output
            const uint32_t gb = (uint32_t)grid_for(ctx, a.Pb);
            hipLaunchKernelGGL(k_path_finish, dim3(gb), dim3(DRT_BLOCK), 0, ctx->stream, pa, (const double*)gpix, s.d_out_gimg, gb,
                               (const double*)nullptr, 0, 0, DRT_FAST_PARAMS * 3, (double*)nullptr, 0u, (const uint32_t*)counts,
                               0u, s.totals);
        }
        s.path_finished = true;
        return DRT_OK;
    }
    hipLaunchKernelGGL(k_sum_counts, dim3(64), dim3(DRT_BLOCK), 0, ctx->stream, counts, (uint32_t)(2 * n_waves),
                       s.totals, (uint32_t)n_waves, 0ull, 0ull, 1u);
    if (backward) {
        DRT_TIMED(s, DRT_K_GRADREDUCE,
                  hipLaunchKernelGGL(k_gradreduce, dim3(gen ? ctx->n_params * 3 : (s.n_fast > 0 ? s.n_fast * 3 : 1)), dim3(DRT_BLOCK), 0, ctx->stream, gpart,
                                     gpath, g_rows, s.grad, g_stride, slot_map));
        st->units[DRT_K_GRADREDUCE] += (uint64_t)gpath;
    }
    if (s.film) {
        DRT_TIMED(s, DRT_K_FILM,
                  hipLaunchKernelGGL(k_film_parts, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, fpart, pa.n_ranges,
                                     a.Pb, a.p0, s.film));
        st->units[DRT_K_FILM] += a.n_paths;
    }
    if (gpix && s.gfilm)
        hipLaunchKernelGGL(k_film_parts, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, gpix, pa.n_ranges, a.Pb, a.p0,
                           s.gfilm);
    return DRT_OK;
}

// ---- route 2: the queue wavefront ----------------------------------------------------------------------------------------
// One pass over the depths [first, D) of the batch's queues: per depth (or per `bounces_from` depths, fused) the BVH walk
// where the scene has a mesh, then K3.  Used for the camera paths (first = 0) and for the suffix paths of the unbiased
// operator's adjoint rounds (first = the round's depth, seg = first: draw positions come from the chain, cs.dbase).
//   parity0       which ping-pong queue buffer holds depth `first`
//   camera_fused  the depth-0 launch generates its camera rays itself (K1 folded in)
//   sv_*          where the launch of depth `first` saves its ray and FINAL hit (the unbiased operator's chain vertices)
//   read / written  bit k - base set: a shade launch started from / ended on the queue of depth k (k_sum_counts adds them up)
template <typename R>
int bounce_loop(Shard<R>& s, int first, int parity0, bool camera_fused, typename Shard<R>::R4* sv_a0, typename Shard<R>::R2* sv_b0,
                HitRec<R>* sv_hit0, int seg, const uint32_t* dbase, int base, unsigned long long* read, unsigned long long* written)
{
    typedef typename Shard<R>::R4 R4;
    typedef typename Shard<R>::R2 R2;
    drt_hip_ctx* ctx = s.ctx;
    const BatchArgs& a = s.a;
    const int D = s.D;
    const bool fused = s.can_fuse, shade_tail = s.shade_tail;
    const int g = (int)((a.n_regions + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE));       // one wave per region
    const int gk2 = grid_for(ctx, a.n_paths);
    int rc;
    for (int k = first, lc = 0, nbk = 1, next_poll = first + DRT_POLL_EVERY; k < D; k += nbk, ++lc) {
        // (queue lanes: by the launch's parity; scenes with a mesh: by the depth mod s.tail_ring -- a two-stage launch appends to two queues)
        const int ringm = s.tail_ring;
        const int cur = shade_tail ? k % ringm : (parity0 + lc) & 1, nxt = shade_tail ? (k + 1) % ringm : cur ^ 1;
        nbk = s.bounces_from(k);
        if (!(camera_fused && k == 0)) *read |= 1ull << (k - base);         // (the camera launch generates its rays)
        if (k + nbk < D) *written |= 1ull << (k + nbk - base);
        uint32_t* ck = s.counts + (size_t)k * s.max_regions;
        if (D > 2 * DRT_POLL_EVERY && k >= next_poll) {
            // (deep caps = roulette-terminated renders: ask every few bounces whether any path is still alive)
            next_poll = k + DRT_POLL_EVERY;
            unsigned long long live = 0;
            // (scenes with a mesh: a ray kept in registers through depth k - 1 waits in the queue of depth k + 1)
            if ((rc = queue_length(ctx, ck, (shade_tail && k + 1 < D ? 2u : 1u) * s.max_regions, &live)) != DRT_OK) return rc;
            if (live == 0)
                break;        // every path has ended: deeper queues stay empty
        }
        // (fused or not, the shade launch has the ray and its final hit in registers)
        R4* sv_a = k == first ? sv_a0 : (R4*)nullptr;
        R2* sv_b = k == first ? sv_b0 : (R2*)nullptr;
        HitRec<R>* sv_hit = k == first ? sv_hit0 : (HitRec<R>*)nullptr;
        // hit lane of this depth (double-buffered where the shade launch fills the next depth's itself: mesh scenes)
        HitRec<R>* hit_k = shade_tail ? s.hitr[k % ringm] : s.hit;
        if (!fused && !shade_tail) {
            // K2 as a kernel of its own (DRT_RENDER_UNFUSED: the textbook wavefront)
            DRT_TIMED(s, DRT_K_INTERSECT,
                      hipLaunchKernelGGL(k_intersect<R>, dim3(gk2), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.ra[cur], s.rb[cur],
                                         hit_k, ck, s.totals));
        }
        if (ctx->has_mesh) {
            // the BVH walk continues from the analytic hit: (t, primitive) refined.  Its candidate lists (one per queue region,
            // pulled shade_list_group at a time) come from the launch that PRODUCED this depth's rays.
            const int gm = (int)std::min<uint64_t>(((uint64_t)a.n_paths + DRT_BLOCK - 1) / DRT_BLOCK, (uint64_t)ctx->n_cu * ctx->mesh_blocks_per_cu);
            const uint32_t walk_group = (uint32_t)tuning().shade_list_group;
            DRT_TIMED(s, DRT_K_INTERSECT_MESH,
                      hipLaunchKernelGGL(k_intersect_mesh<R>, dim3(gm), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.bvh, hit_k,
                                         (const uint32_t*)ctx->cand[s.cset(k)].p, (const R4*)ctx->cand_a[s.cset(k)].p, (const R4*)ctx->cand_b[s.cset(k)].p,
                                         (uint32_t*)ctx->cand_count[s.cset(k)].p, s.region_size, a.n_regions, walk_group,
                                         coprime_multiplier((a.n_regions + walk_group - 1) / walk_group), s.totals));
        }
        TapeRec<R>* tape_k = s.tape + (size_t)k * a.n_paths;
        const bool tail_here = shade_tail && k + nbk < D;
        // the launch's two stages: the vertex of depth k, and -- for the rays that miss the bounds of the mesh, whose analytic
        // hit is final -- the vertex of depth k + 1 in registers.  The lists of depth k + 1 hold what launch k - 1's second
        // stage left there; the lists of depth k + 2 (= the ones the walk of depth k has just consumed) start empty.
        const int tail_nb = tail_here ? s.tail_nb : 1;
        if (tail_here) {     // (the region lists of regions no wave visits stay empty)
            if (tail_nb == 1 || lc == 0)
                HIPCHK(ctx, hipMemsetAsync(ctx->cand_count[s.cset(k + 1)].p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
            if (tail_nb > 1)
                HIPCHK(ctx, hipMemsetAsync(ctx->cand_count[s.cset(k)].p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
        }
        const TailQueue<R> tq0 = s.tailq(k + 1), tq1 = s.tailq(k + 2);
        uint32_t* cont_row = s.counts + (size_t)(D + 1) * s.max_regions;
#define DRT_SHADE_ARGS a, k, nbk, s.d_scene, s.d_params, s.ra[cur], s.rb[cur], s.rid[cur], hit_k, s.ra[nxt], s.rb[nxt], s.rid[nxt], tape_k, \
                       s.nv, ck, (uint32_t)s.max_regions, s.bvh.tri_shade
#define DRT_SHADE_ARGS_TAIL a, k, tail_nb, s.d_scene, s.d_params, s.ra[cur], s.rb[cur], s.rid[cur], hit_k, s.ra[nxt], s.rb[nxt], s.rid[nxt], tape_k, \
                            s.nv, ck, (uint32_t)s.max_regions, s.bvh.tri_shade
#define DRT_SHADE_NO_TAIL s.bvh, tq0, tq1, (uint32_t*)nullptr
#define DRT_SHADE_TAIL s.bvh, tq0, tq1, cont_row
#define DRT_LAUNCH_SHADE(SPEC)                                                                                                       \
    do {                                                                                                                            \
        if (fused && camera_fused && k == 0)                                                                                        \
            hipLaunchKernelGGL((k_shade<R, SPEC, true, true>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, DRT_SHADE_ARGS, 0,          \
                               (const uint32_t*)nullptr, sv_a, sv_b, sv_hit, DRT_SHADE_NO_TAIL);                                    \
        else if (fused)                                                                                                             \
            hipLaunchKernelGGL((k_shade<R, SPEC, true>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, DRT_SHADE_ARGS, seg, dbase,       \
                               sv_a, sv_b, sv_hit, DRT_SHADE_NO_TAIL);                                                              \
        else if (tail_here)                                                                                                         \
            hipLaunchKernelGGL((k_shade<R, SPEC, false, false, true>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, DRT_SHADE_ARGS_TAIL, \
                               seg, dbase, sv_a, sv_b, sv_hit, DRT_SHADE_TAIL);                                                     \
        else                                                                                                                        \
            hipLaunchKernelGGL((k_shade<R, SPEC, false>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, DRT_SHADE_ARGS, seg, dbase,      \
                               sv_a, sv_b, sv_hit, DRT_SHADE_NO_TAIL);                                                              \
    } while (0)
        if (ctx->has_specular)
            DRT_TIMED(s, DRT_K_SHADE, DRT_LAUNCH_SHADE(true));
        else
            DRT_TIMED(s, DRT_K_SHADE, DRT_LAUNCH_SHADE(false));
#undef DRT_LAUNCH_SHADE
#undef DRT_SHADE_TAIL
#undef DRT_SHADE_NO_TAIL
#undef DRT_SHADE_ARGS_TAIL
#undef DRT_SHADE_ARGS
    }
    return DRT_OK;
}

// the unbiased operator's adjoint rounds (integrate.hpp:11-24, 39-52): round r re-samples every chain vertex (k_adj_vertex:
// fresh theta / phi, the suffix's first ray queued at depth r + 1), traces the suffix with the ordinary bounce loop, and
// walks its tape for L', adds the round's gradients and moves the chain on (k_adj_accumulate)
template <typename R>
int adjoint_rounds(Shard<R>& s)
{
    typedef typename Shard<R>::R4 R4;
    drt_hip_ctx* ctx = s.ctx;
    const BatchArgs& a = s.a;
    const int D = s.D;
    const int g = (int)((a.n_regions + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE));
    const int gp = grid_for(ctx, a.n_paths);
    int rc;
    // forward radiance from the tape, then the rounds
    if ((rc = timing_begin(ctx, s.timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
    if (s.film)
        hipLaunchKernelGGL(k_radiance<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.d_params, s.tape, s.nv, s.lacc);
    hipLaunchKernelGGL(k_adj_init<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, s.tape, s.nv, s.d_adjoint, s.cs);
    if ((rc = timing_end(ctx, s.timing)) != DRT_OK) return rc;
    s.st->launches[DRT_K_BACKWARD]++;
    for (int r = 0; r < D; ++r) {
        const int sd = r + 1;                                    // depth of the suffix's first ray
        HIPCHK(ctx, hipMemsetAsync(s.counts + (size_t)sd * s.max_regions, 0, (size_t)(D + 2 - sd) * s.max_regions * sizeof(uint32_t), ctx->stream));
        // (scenes with a mesh: the kernel also intersects the rays it queues with the analytic shapes and builds the BVH walk's
        //  candidate lists -- hit lane `hit`, the one the suffix loop starts on.  Timed with the backward pass.)
        if (s.shade_tail)
            HIPCHK(ctx, hipMemsetAsync(ctx->cand_count[s.cset(sd)].p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
        const int sq = s.shade_tail ? sd % s.tail_ring : sd & 1;   // the queue lanes of depth sd
#define DRT_LAUNCH_ADJ_VERTEX(SPEC, TAILV)                                                                                  \
    hipLaunchKernelGGL((k_adj_vertex<R, SPEC, TAILV>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, a, r, s.d_scene, s.d_params, s.cs, \
                       s.bvh.tri_shade, s.ra[sq], s.rb[sq], s.rid[sq], s.nv, s.counts + (size_t)sd * s.max_regions, s.bvh, \
                       s.shade_tail ? s.hitr[sq] : s.hit, (uint32_t*)ctx->cand[s.cset(sd)].p, (R4*)ctx->cand_a[s.cset(sd)].p, (R4*)ctx->cand_b[s.cset(sd)].p, (uint32_t*)ctx->cand_count[s.cset(sd)].p)
        if (s.shade_tail) {
            if (ctx->has_specular) DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_VERTEX(true, true));
            else DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_VERTEX(false, true));
        } else {
            if (ctx->has_specular) DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_VERTEX(true, false));
            else DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_VERTEX(false, false));
        }
#undef DRT_LAUNCH_ADJ_VERTEX
        bool chains_done = false;
        if (D > 2 * DRT_POLL_EVERY && r >= 2) {
            // no suffix ray queued in this round => every chain ends with this round
            unsigned long long live = 0;
            if ((rc = queue_length(ctx, s.counts + (size_t)sd * s.max_regions, s.max_regions, &live)) != DRT_OK) return rc;
            chains_done = live == 0;
        }
        // the suffix: every depth gets its analytic hit and candidate lists from the launch that PRODUCES its rays
        // (k_adj_vertex<TAIL> for depth sd, k_shade<TAIL> after it); the suffix's first ray and its FINAL hit -- the next chain
        // vertex -- are saved by the shade launch of depth sd, which holds both
        unsigned long long sfx_read = 0, sfx_written = 0;     // rows relative to depth sd
        if (!chains_done)
            if ((rc = bounce_loop<R>(s, sd, sd & 1, false, s.cs.nx_a, s.cs.nx_b, s.cs.nx_hit, sd, (const uint32_t*)s.cs.dbase, sd,
                                     &sfx_read, &sfx_written)) != DRT_OK) return rc;
        // (rows sd .. D - 1: the suffix's segments; row D: the suffixes a user cap cut short -- capped_paths counts every walk
        //  the cap ended, the camera paths' and the suffixes', like the one-launch kernel does)
        if (sd < D)
            hipLaunchKernelGGL(k_sum_counts, dim3(64), dim3(DRT_BLOCK), 0, ctx->stream, s.counts + (size_t)sd * s.max_regions,
                               (uint32_t)((size_t)(D + 2 - sd) * s.max_regions), s.totals, (uint32_t)s.max_regions, sfx_read, sfx_written,
                               (uint32_t)(D - sd));
#define DRT_LAUNCH_ADJ_ACC(NP)                                                                                           \
    hipLaunchKernelGGL((k_adj_accumulate<R, NP>), dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, r, s.d_scene, s.d_params, \
                       s.tape, s.nv, s.cs, s.gpart, s.grad, s.g_rows, s.g_stride)
        if (ctx->n_params <= 4) DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_ACC(4));
        else if (ctx->n_params <= 8) DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_ACC(8));
        else DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_ADJ_ACC(0));
#undef DRT_LAUNCH_ADJ_ACC
        if (tuning().dump_path >= -2 && tuning().dump_path != -1) {       // debugging aid (DRT_HIP_DUMP_PATH = <path of the batch> | -2: every path)
            (void)hipStreamSynchronize(ctx->stream);
            const size_t first = tuning().dump_path >= 0 ? (size_t)tuning().dump_path : 0;
            const size_t last = tuning().dump_path >= 0 ? first + 1 : (a.n_paths <= 8192 ? (size_t)a.n_paths : 0);
            for (size_t i = first; i < last && i < a.n_paths; ++i) {
                uint32_t k_nv = 0, nd = 0, db = 0;
                HitRec<R> hn, hc;
                R4 gg;
                (void)hipMemcpy(&hc, s.cs.cv_hit + i, sizeof hc, hipMemcpyDeviceToHost);
                if (hc.prim < 0)
                    continue;                                   // (the chain of this path has ended)
                (void)hipMemcpy(&k_nv, s.nv + i, sizeof k_nv, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&nd, s.cs.ndraw + i, sizeof nd, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&db, s.cs.dbase + i, sizeof db, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&hn, s.cs.nx_hit + i, sizeof hn, hipMemcpyDeviceToHost);
                (void)hipMemcpy(&gg, s.cs.g + i, sizeof gg, hipMemcpyDeviceToHost);
                const uint32_t ids_ = 0;
                (void)ids_;
                fprintf(stderr, "[drt_hip] path %zu (pixel %u sample %u) round %d: chain prim %d, suffix base %u, %d suffix vertices, draws after %u, next prim %d\n",
                        i, (unsigned)(i % a.Pb), (unsigned)(i / a.Pb), r, hc.prim, db, (int)k_nv - (r + 1), nd, hn.prim);
            }
        }
        std::swap(s.cs.cv_a, s.cs.nx_a);                // the suffix's first vertex is the chain's next one
        std::swap(s.cs.cv_b, s.cs.nx_b);
        std::swap(s.cs.cv_hit, s.cs.nx_hit);
        DRT_TIMED(s, DRT_K_GRADREDUCE,
                  hipLaunchKernelGGL(k_gradreduce, dim3(s.g_rows > 0 ? s.g_rows : 1), dim3(DRT_BLOCK), 0, ctx->stream, s.gpart, gp,
                                     s.g_rows, s.grad, s.g_stride));
        if (chains_done)
            break;
    }
    return DRT_OK;
}

template <typename R>
int queue_batch(Shard<R>& s)
{
    typedef typename Shard<R>::R4 R4;
    drt_hip_ctx* ctx = s.ctx;
    const BatchArgs& a = s.a;
    drt_hip_stats* st = s.st;
    const int D = s.D;
    int rc;
    HIPCHK(ctx, hipMemsetAsync(s.counts, 0, s.cw * sizeof(uint32_t), ctx->stream));
    const int g = (int)((a.n_regions + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE));
    const int gp = grid_for(ctx, a.n_paths);   // per-path kernels (K6): persistent grid
    // K1 folded into the first shade launch when it is a fused one that carries its rays through several bounces and every
    // path is alive at depth 0 (with one launch per bounce the depth-0 launch is the largest, and the camera's f64 math no
    // longer hides behind K1's own writes)
    const bool camera_fused = s.can_fuse && D > 0 && a.min_bounces > 0 && s.bounces_from(0) > 1;
    if (!camera_fused) {
        // (scenes with a mesh: K1 also intersects its rays with the analytic shapes and builds the BVH walk's candidate lists --
        //  hit lane `hit`, the one the bounce loop starts on)
        if (s.shade_tail) {
            HIPCHK(ctx, hipMemsetAsync(ctx->cand_count[0].p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
            DRT_TIMED(s, DRT_K_RAYGEN,
                      hipLaunchKernelGGL((k_raygen<R, true>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.ra[0], s.rb[0],
                                         s.rid[0], s.nv, s.counts, s.bvh, s.hit, (uint32_t*)ctx->cand[0].p, (R4*)ctx->cand_a[0].p,
                                         (R4*)ctx->cand_b[0].p, (uint32_t*)ctx->cand_count[0].p));
        } else
            DRT_TIMED(s, DRT_K_RAYGEN,
                      hipLaunchKernelGGL((k_raygen<R, false>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.ra[0], s.rb[0],
                                         s.rid[0], s.nv, s.counts, s.bvh, (HitRec<R>*)nullptr, (uint32_t*)nullptr, (R4*)nullptr,
                                         (R4*)nullptr, (uint32_t*)nullptr));
        st->units[DRT_K_RAYGEN] += a.n_paths;
    }
    // the camera paths.  Unbiased: the camera ray's hit is the first chain vertex of the backward pass, saved by the depth-0 launch
    unsigned long long read_rows = 0, written_rows = 0;
    if ((rc = bounce_loop<R>(s, 0, 0, camera_fused, s.unbiased ? s.cs.cv_a : (R4*)nullptr,
                             s.unbiased ? s.cs.cv_b : (typename Shard<R>::R2*)nullptr, s.unbiased ? s.cs.cv_hit : (HitRec<R>*)nullptr, 0,
                             (const uint32_t*)nullptr, 0, &read_rows, &written_rows)) != DRT_OK) return rc;
    hipLaunchKernelGGL(k_sum_counts, dim3(64), dim3(DRT_BLOCK), 0, ctx->stream, s.counts, (uint32_t)((size_t)(D + 2) * s.max_regions),
                       s.totals, (uint32_t)s.max_regions, read_rows, written_rows, (uint32_t)D);
    if (s.backward && D > 0 && s.gimg_param >= 0) {
        // gradient image: per-path gradient of one parameter, averaged per pixel by K5
        DRT_TIMED(s, DRT_K_BACKWARD,
                  hipLaunchKernelGGL(k_backward_image<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.d_params, s.tape,
                                     s.nv, s.d_adjoint, (uint32_t)s.gimg_param, (R4*)ctx->gpath.p, s.film ? s.lacc : (R4*)nullptr));
        hipLaunchKernelGGL(k_film<R>, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, a, (const R4*)ctx->gpath.p, s.gfilm);
    } else if (s.unbiased && D > 0) {
        if ((rc = adjoint_rounds<R>(s)) != DRT_OK) return rc;
    } else if (s.backward && D > 0) {
        // DRT_RENDER_LOSS_L2: the seed of a path is 2 (L_path - target): the radiance walk first, then the gradient walk
        if (s.loss_l2)
            DRT_TIMED(s, DRT_K_BACKWARD,
                      hipLaunchKernelGGL(k_radiance<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.d_params, s.tape, s.nv, s.lacc));
#define DRT_LAUNCH_BWD(NP)                                                                                       \
    hipLaunchKernelGGL((k_backward<R, NP>), dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.d_params, \
                       s.tape, s.nv, s.d_adjoint, s.gpart, s.grad, (s.film && !s.loss_l2) ? s.lacc : (R4*)nullptr, s.g_rows, \
                       s.g_stride, s.loss_l2 ? (const R4*)s.lacc : (const R4*)nullptr)
        if (ctx->n_params <= 4) DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_BWD(4));
        else if (ctx->n_params <= 8) DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_BWD(8));
        else DRT_TIMED(s, DRT_K_BACKWARD, DRT_LAUNCH_BWD(0));
#undef DRT_LAUNCH_BWD
        DRT_TIMED(s, DRT_K_GRADREDUCE,
                  hipLaunchKernelGGL(k_gradreduce, dim3(s.g_rows > 0 ? s.g_rows : 1), dim3(DRT_BLOCK), 0, ctx->stream, s.gpart, gp,
                                     s.g_rows, s.grad, s.g_stride));
        st->units[DRT_K_GRADREDUCE] += (uint64_t)gp;
    } else if (D > 0 && s.film) {
        // forward only: radiance of every path from its tape
        DRT_TIMED(s, DRT_K_BACKWARD,
                  hipLaunchKernelGGL(k_radiance<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, s.d_scene, s.d_params, s.tape, s.nv, s.lacc));
    }
    if (D <= 0 && s.film)
        HIPCHK(ctx, hipMemsetAsync(s.lacc, 0, (size_t)a.n_paths * sizeof(R4), ctx->stream));
    if (s.film) {
        DRT_TIMED(s, DRT_K_FILM, hipLaunchKernelGGL(k_film<R>, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, a, s.lacc, s.film));
        st->units[DRT_K_FILM] += a.n_paths;
    }
    return DRT_OK;
}

// ---- one shard's render, enqueued ------------------------------------------------------------------------------------------
template <typename R>
int render_impl(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                const float* d_adjoint, float* d_out_rgb, bool backward, bool timing,
                drt_hip_stats* st, uint32_t n_local_pixels, int depth_cap, size_t* n_count_words,
                double* film, int gimg_param = -1, double* gfilm = nullptr, float* d_out_gimg = nullptr)
{
    Shard<R> s;
    s.ctx = ctx; s.cam = cam; s.rp = rp; s.d_adjoint = d_adjoint; s.d_out_rgb = d_out_rgb; s.backward = backward; s.timing = timing;
    s.st = st; s.n_local_pixels = n_local_pixels; s.D = depth_cap; s.film = film; s.gimg_param = gimg_param; s.gfilm = gfilm;
    s.d_out_gimg = d_out_gimg;
    s.d_scene = sizeof(R) == 4 ? (const DevScene<R>*)ctx->d_scene_f : (const DevScene<R>*)ctx->d_scene_d;
    s.d_params = sizeof(R) == 4 ? (const R*)ctx->d_params_f : (const R*)ctx->d_params_d;
    memcpy(&s.bvh, sizeof(R) == 4 ? (const void*)&ctx->bvh_f : (const void*)&ctx->bvh_d, sizeof s.bvh);
    shard_plan(s);
    if (!ctx->user_header.empty()) {
        // caller-defined shape kinds live in the one-launch path kernel hiprtc compiles for the scene, nowhere else
        if (ctx->jit_mode <= 0)
            return fail(ctx, DRT_ERR_UNSUPPORTED, "render: the scene holds caller-defined shape kinds (DRT_SHAPE_USER) and this context may not compile "
                                                  "(drt_hip_set_specialisation / DRT_HIP_JIT)");
        if (!s.use_path || s.mesh_path)
            return fail(ctx, DRT_ERR_UNSUPPORTED, "render: caller-defined shape kinds render on the one-launch path kernels only -- not with "
                                                  "bounces_per_launch >= 1, DRT_RENDER_UNFUSED, a gradient image of more than 8 parameters, or more "
                                                  "parameters than the kernels stage (136)");
    }
    int rc;
    if (ctx->adj_pending && s.d_adjoint) {
        // a host-buffer render's adjoint image sits in pinned memory: the one-launch routes read a pixel's seed ONCE per lane, so
        // a small frame takes it from there; everything else (the tape route reads it per path) gets the copy in device memory
        ctx->adj_pending = false;
        if (s.use_path && (size_t)n_local_pixels * 3 * sizeof(float) <= ((size_t)1 << 20))
            s.d_adjoint = (const float*)ctx->adj_src_dev;
        else
            HIPCHK(ctx, hipMemcpyAsync(ctx->adjoint.p, ctx->adj_src_host, ctx->adj_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = shard_buffers(s)) != DRT_OK) return rc;
    *n_count_words = s.cw;
    if (!s.path_finish) {
        HIPCHK(ctx, hipMemsetAsync(s.totals, 0, DRT_TOTAL_WORDS * sizeof(unsigned long long), ctx->stream));
        if (film)
            HIPCHK(ctx, hipMemsetAsync(film, 0, (size_t)n_local_pixels * 3 * sizeof(double), ctx->stream));
        if (backward)
            HIPCHK(ctx, hipMemsetAsync(s.grad, 0, (size_t)(ctx->n_params ? ctx->n_params : 1) * 3 * sizeof(double), ctx->stream));
    }
    shard_args(s);
    BatchArgs& a = s.a;
    uint64_t batch = 0;
    for (uint32_t p0 = 0; p0 < n_local_pixels; p0 += s.Pb) {
        for (uint32_t s0 = 0; s0 < (uint32_t)s.spp; s0 += s.Sb, ++batch) {
            a.p0 = p0; a.s0 = s0;
            a.Pb = (n_local_pixels - p0) < s.Pb ? (n_local_pixels - p0) : s.Pb;
            a.Sb = ((uint32_t)s.spp - s0) < s.Sb ? ((uint32_t)s.spp - s0) : s.Sb;
            a.n_paths = a.Pb * a.Sb;
            a.n_regions = (a.n_paths + s.region_size - 1) / s.region_size;
            if ((rc = s.use_path ? path_batch<R>(s) : queue_batch<R>(s)) != DRT_OK) return rc;
        }
    }
    // debugging aid: DRT_HIP_DUMP_PATH=<path index in the last batch> prints that path's tape
    if (tuning().dump_path >= 0 && !s.use_path) {
        const size_t i = (size_t)tuning().dump_path;
        if (i < a.n_paths && s.D > 0) {
            (void)hipStreamSynchronize(ctx->stream);
            uint32_t k_nv = 0;
            (void)hipMemcpy(&k_nv, s.nv + i, sizeof k_nv, hipMemcpyDeviceToHost);
            fprintf(stderr, "[drt_hip] path %zu: %u vertices\n", i, k_nv);
            for (uint32_t k = 0; k < k_nv && k < (uint32_t)s.D; ++k) {
                TapeRec<R> tr;
                (void)hipMemcpy(&tr, s.tape + (size_t)k * a.n_paths + i, sizeof tr, hipMemcpyDeviceToHost);
                fprintf(stderr, "[drt_hip]   k=%u m=%.9g colour=%u emission=%u\n", k, (double)tr.m, tr.ids & 0xFFFFu, tr.ids >> 16);
            }
        }
    }
    st->batches = batch;
    st->paths = s.total_paths;
    if (film && d_out_rgb && !s.path_finished)
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, n_local_pixels)), dim3(DRT_BLOCK), 0, ctx->stream, a, n_local_pixels, film, d_out_rgb);
    if (gimg_param >= 0 && gfilm && d_out_gimg && !s.path_finished)
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, n_local_pixels)), dim3(DRT_BLOCK), 0, ctx->stream, a, n_local_pixels, gfilm, d_out_gimg);
    {   // this render's lane of partial-sum buffers is free once the context's stream has come this far
        const int lane = s.overlap_ok ? (ctx->slot & 1) : 0;
        if (ctx->ev_lane_free[lane]) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_lane_free[lane], ctx->stream));
            ctx->lane_used[lane] = true;
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return DRT_OK;
}

} // namespace
