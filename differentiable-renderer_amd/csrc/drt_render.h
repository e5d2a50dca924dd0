// drt_render.h -- a render call in phases (launch -> reduce -> collect -> finish), for a plain context, a group context (every
// phase on all members before the next) and asynchronous frames; the ONE collective of the path (ncclAllReduce of the P x 3
// gradient accumulator, VariableNode::backward's `m_grad += grad`, vector.hpp:185-188) is enqueued in `reduce`.
#pragma once
#include <thread>

// rows [y0, y1) of the bands that `shard` owns
template <typename F>
static void for_each_band(int height, int band, int n_shards, int shard, F&& fn)
{
    if (n_shards <= 1) {
        fn(0, height);
        return;
    }
    for (int y0 = shard * band; y0 < height; y0 += n_shards * band)
        fn(y0, y0 + band < height ? y0 + band : height);
}

static int ensure_stage(drt_hip_ctx* ctx, RenderJob& j);

// the device's view of [p, p + bytes) if the caller pinned a range that contains it (drt_hip_pin_host), else nullptr
static uint8_t* pinned_alias(const drt_hip_ctx* ctx, const void* p, size_t bytes)
{
    const uint8_t* q = (const uint8_t*)p;
    if (!q)
        return nullptr;
    for (const drt_hip_ctx::PinnedRange& r : ctx->pinned)
        if (q >= r.host && q + bytes <= r.host + r.bytes)
            return r.dev + (q - r.host);
    return nullptr;
}

// phase 1: validate, set up, enqueue the whole pipeline; the gradient of THIS context's shard ends up in ctx->grad[ctx->slot]
static int render_launch(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                         const float* adjoint_rgb, float* out_rgb, double* out_param_grad, drt_hip_stats* stats,
                         int gimg_param, float* out_gimg)
{
    if (!ctx->has_scene)
        return fail(ctx, DRT_ERR_NO_SCENE, "render before upload_scene");
    if (!cam || !rp || cam->width <= 0 || cam->height <= 0 || rp->spp <= 0 || rp->min_bounces < 0 ||
        !(rp->absorb >= 0.0 && rp->absorb <= 1.0))
        return fail(ctx, DRT_ERR_INVALID, "render: bad camera or render parameters");
    if ((uint64_t)cam->width * (uint64_t)cam->height >= 0xFFFFFFFFull)
        return fail(ctx, DRT_ERR_INVALID, "render: image too large");
    // the path index (pixel * spp + sample) of every camera sample of the FRAME fits 32 bits: the kernels keep its low word
    // as the path's RNG key and share the high word's hash round (drt_hip.h: path_hi = 0)
    if ((uint64_t)cam->width * (uint64_t)cam->height * (uint64_t)rp->spp > (1ull << 32))
        return fail(ctx, DRT_ERR_INVALID, "render: more than 2^32 camera samples in one frame (width x height x spp)");
    if (rp->max_depth > DRT_MAX_DEPTH)
        return fail(ctx, DRT_ERR_INVALID, "render: max_depth above DRT_MAX_DEPTH (64)");
    if (rp->absorb >= 1.0 && rp->max_depth <= 0 && rp->min_bounces > DRT_MAX_DEPTH)
        return fail(ctx, DRT_ERR_INVALID, "render: absorb == 1 ends every path at min_bounces, which is above DRT_MAX_DEPTH (64)");
    const int n_shards = rp->n_shards > 1 ? rp->n_shards : 1;
    const int band = rp->band_rows > 0 ? rp->band_rows : 1;
    if (n_shards > 1 && (rp->shard < 0 || rp->shard >= n_shards))
        return fail(ctx, DRT_ERR_INVALID, "render: shard out of range");
    RenderJob& j = ctx->job;
    j = RenderJob();
    j.cam = *cam;
    j.rp = *rp;
    j.adjoint_rgb = adjoint_rgb; j.out_rgb = out_rgb; j.out_param_grad = out_param_grad; j.out_gimg = out_gimg;
    j.stats = stats;
    j.gimg_param = gimg_param;
    j.n_shards = n_shards; j.shard = n_shards > 1 ? rp->shard : 0; j.band = band;
    j.backward = (rp->flags & DRT_RENDER_BACKWARD) != 0;
    j.dev_out = (rp->flags & DRT_RENDER_DEVICE_OUT) != 0;
    j.timing = (rp->flags & DRT_RENDER_TIMING) != 0;
    const bool f64 = (rp->flags & DRT_RENDER_F64) != 0;
    if (j.backward && !out_param_grad && gimg_param < 0 && !ctx->is_member)
        return fail(ctx, DRT_ERR_INVALID, "render: DRT_RENDER_BACKWARD needs out_param_grad");
    if ((rp->flags & DRT_RENDER_LOSS_L2) && j.backward &&
        (!adjoint_rgb || (rp->flags & DRT_RENDER_UNBIASED) || gimg_param >= 0))
        return fail(ctx, DRT_ERR_INVALID, "render: DRT_RENDER_LOSS_L2 needs a target image in adjoint_rgb, the biased operator and summed gradients");
    if ((rp->flags & (DRT_RENDER_ALLREDUCE | DRT_RENDER_ALLREDUCE_ASYNC)) && j.backward && !ctx->comm && !ctx->is_member)
        return fail(ctx, DRT_ERR_INVALID, "render: DRT_RENDER_ALLREDUCE on a context without a communicator (drt_hip_comm_init_rank)");

    j.t0 = std::chrono::steady_clock::now();
    HIPCHK(ctx, hipSetDevice(ctx->device));

    // rows owned by this shard
    uint32_t local_rows = 0;
    for_each_band(cam->height, band, n_shards, j.shard, [&](int y0, int y1) { local_rows += (uint32_t)(y1 - y0); });
    j.n_local_pixels = local_rows * (uint32_t)cam->width;

    // deepest vertex a path can reach: absorb == 1 kills every path at depth min_bounces
    int depth_cap = rp->max_depth > 0 ? rp->max_depth : DRT_MAX_DEPTH;
    if (rp->absorb >= 1.0 && rp->min_bounces < depth_cap)
        depth_cap = rp->min_bounces;

    memset(&j.st, 0, sizeof j.st);
    ctx->events_used = 0;
    ctx->timed.clear();

    int rc;
    const size_t npix_all = (size_t)cam->width * cam->height;
    const float* d_adj = nullptr;
    if (out_rgb) {
        if ((rc = ensure(ctx, ctx->film, (size_t)(j.n_local_pixels ? j.n_local_pixels : 1) * 3 * sizeof(double))) != DRT_OK) return rc;
        uint8_t* pinned_out = j.dev_out ? nullptr : pinned_alias(ctx, out_rgb, npix_all * 3 * sizeof(float));
        if (j.dev_out) {
            j.d_out = out_rgb;
        } else if (pinned_out) {
            // (the caller pinned this buffer, drt_hip_pin_host: the finishing kernels write the image straight into it)
            j.direct_out = true;
            if ((rc = ensure_stage(ctx, j)) != DRT_OK) return rc;
            j.d_out = (float*)pinned_out;
        } else {
            if (ctx->zero_copy_next) {
                // (the finishing kernels write the image into the pinned block of this frame: no device image, no copy)
                j.zero_copy = true;
                if ((rc = ensure_stage(ctx, j)) != DRT_OK) return rc;
                j.d_out = (float*)(ctx->h_stage[ctx->slot] + j.off_img);
            } else {
                if ((rc = ensure(ctx, ctx->out[ctx->slot], npix_all * 3 * sizeof(float))) != DRT_OK) return rc;
                j.d_out = (float*)ctx->out[ctx->slot].p;       // only this shard's rows are written, and only they are copied back
            }
        }
    }
    if (j.backward) {
        if ((rc = ensure(ctx, ctx->grad[ctx->slot], (size_t)(ctx->n_params ? ctx->n_params : 1) * 3 * sizeof(double))) != DRT_OK) return rc;
        if (adjoint_rgb) {
            if (j.dev_out) {
                d_adj = adjoint_rgb;
            } else {
                // The adjoint image in pinned memory: the caller's own buffer if he pinned it (drt_hip_pin_host), else a copy in the
                // context's pinned block.  render_impl decides how the kernels get it (adjoint_to_device): small frames on the
                // one-launch route read it from there, one 12-byte load per pixel -- no DMA in front of the frame, ~20 us of a
                // small frame --, everything else gets it copied into device memory (the tape route reads it per PATH).
                if ((rc = ensure(ctx, ctx->adjoint, npix_all * 3 * sizeof(float))) != DRT_OK) return rc;
                ctx->adj_src_host = adjoint_rgb;
                // (an ASYNCHRONOUS frame always takes its copy: drt_hip_render_async promises that adjoint_rgb is consumed when it
                //  returns -- a loop rewrites its one pinned adjoint image for the next frame while this one is in flight)
                ctx->adj_src_dev = ctx->stage_adjoint_next ? nullptr : pinned_alias(ctx, adjoint_rgb, npix_all * 3 * sizeof(float));
                if (!ctx->adj_src_dev) {
                    if ((rc = ensure_stage(ctx, j)) != DRT_OK) return rc;
                    memcpy(ctx->h_stage[ctx->slot] + j.off_adj, adjoint_rgb, npix_all * 3 * sizeof(float));
                    ctx->adj_src_host = ctx->adj_src_dev = ctx->h_stage[ctx->slot] + j.off_adj;
                }
                ctx->adj_bytes = npix_all * 3 * sizeof(float);
                ctx->adj_pending = true;
                d_adj = (const float*)ctx->adjoint.p;
            }
        }
    }
    if (gimg_param >= 0) {
        const size_t fb = (size_t)(j.n_local_pixels ? j.n_local_pixels : 1) * 3 * sizeof(double);
        if ((rc = ensure(ctx, ctx->gfilm, fb)) != DRT_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->gfilm.p, 0, fb, ctx->stream));
        uint8_t* pinned_gimg = j.dev_out ? nullptr : pinned_alias(ctx, out_gimg, npix_all * 3 * sizeof(float));
        if (j.dev_out) {
            j.d_gimg = out_gimg;
        } else if (pinned_gimg) {
            j.direct_gimg = true;
            j.d_gimg = (float*)pinned_gimg;
        } else {
            if ((rc = ensure(ctx, ctx->gimg_out, npix_all * 3 * sizeof(float))) != DRT_OK) return rc;
            j.d_gimg = (float*)ctx->gimg_out.p;
        }
    }
    double* d_film = out_rgb ? (double*)ctx->film.p : nullptr;   // no image requested: skip K5
    rc = DRT_OK;
    if (j.n_local_pixels == 0 && j.backward)     // (a shard without rows: render_impl, which zeroes the accumulators, is not run)
        HIPCHK(ctx, hipMemsetAsync(ctx->grad[ctx->slot].p, 0, (size_t)(ctx->n_params ? ctx->n_params : 1) * 3 * sizeof(double), ctx->stream));
    if (j.n_local_pixels > 0) {
        if (f64)
            rc = render_impl<double>(ctx, cam, rp, d_adj, j.d_out, j.backward, j.timing, &j.st, j.n_local_pixels,
                                     depth_cap, &j.n_count_words, d_film, gimg_param, (double*)ctx->gfilm.p, j.d_gimg);
        else
            rc = render_impl<float>(ctx, cam, rp, d_adj, j.d_out, j.backward, j.timing, &j.st, j.n_local_pixels,
                                    depth_cap, &j.n_count_words, d_film, gimg_param, (double*)ctx->gfilm.p, j.d_gimg);
    }
    if (ctx->adj_pending) {          // (no kernel wanted it: a shard without rows)
        ctx->adj_pending = false;
    }
    return rc;
}

// phase 2 (one process per GPU): THE collective of the path -- the P x 3 gradient accumulator summed over the ranks
static int render_reduce(drt_hip_ctx* ctx, hipStream_t cs = nullptr)
{
    RenderJob& j = ctx->job;
    if (!(j.backward && (j.rp.flags & (DRT_RENDER_ALLREDUCE | DRT_RENDER_ALLREDUCE_ASYNC)) && ctx->comm && j.gimg_param < 0))
        return DRT_OK;
    const ncclResult_t r = ncclAllReduce(ctx->grad[ctx->slot].p, ctx->grad[ctx->slot].p, (size_t)ctx->n_user_params * 3, ncclDouble, ncclSum,
                                         ctx->comm, cs ? cs : ctx->stream);
    if (r != ncclSuccess) {
        ctx->err = std::string("ncclAllReduce: ") + ncclGetErrorString(r);
        return DRT_ERR_COMM;
    }
    return DRT_OK;
}

// a render that carries an all-reduce failed on THIS rank before the collective was enqueued: the other ranks have
// enqueued theirs (or will) and would wait for this one for ever -- abort the communicator, so that they fail instead
static void abort_comm_after_failure(drt_hip_ctx* ctx, const drt_render_params* rp)
{
    if (ctx->comm && ctx->comm_size > 1 && rp && (rp->flags & DRT_RENDER_BACKWARD) &&
        (rp->flags & (DRT_RENDER_ALLREDUCE | DRT_RENDER_ALLREDUCE_ASYNC))) {
        (void)ncclCommAbort(ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_size = 0;
        ctx->err += " [the communicator was aborted: the other ranks' all-reduce fails instead of hanging]";
    }
}

static int ensure_copy_stream(drt_hip_ctx* ctx)
{
    if (!ctx->copy_stream) {
        // highest priority: the copies (and the all-reduce) of frame i must not queue behind the kernels of frame i + 1,
        // which fill every CU
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, greatest));
    }
    for (int i = 0; i < DRT_HIP_FRAMES_IN_FLIGHT; ++i) {
        if (!ctx->ev_rendered[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_rendered[i], hipEventDisableTiming));
        if (!ctx->ev_copied[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_copied[i], hipEventDisableTiming));
    }
    for (int i = 0; i < 2; ++i) {
        if (!ctx->ev_begin[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_begin[i], hipEventDisableTiming));
        if (!ctx->ev_path[i]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_path[i], hipEventDisableTiming));
    }
    return DRT_OK;
}

// the context's pinned block of one render: [totals 64 B | completion word (a line of its own) | gradients | image | gradient image | adjoint image]
#define DRT_STAGE_DONE 64
static int ensure_stage(drt_hip_ctx* ctx, RenderJob& j)
{
    const size_t npix_all = (size_t)j.cam.width * j.cam.height;
    j.img_bytes = npix_all * 3 * sizeof(float);
    j.grad_bytes = (size_t)ctx->n_user_params * 3 * sizeof(double);
    j.off_grad = 128;
    j.off_img = j.off_grad + ((j.grad_bytes + 15) & ~(size_t)15);
    j.off_gimg = j.off_img + j.img_bytes;
    j.off_adj = j.off_gimg + j.img_bytes;
    const size_t need = j.off_adj + ((j.adjoint_rgb && !j.dev_out) ? j.img_bytes : 0);
    if (ctx->h_stage_cap[ctx->slot] < need) {
        if (ctx->h_stage[ctx->slot])
            (void)hipHostFree(ctx->h_stage[ctx->slot]);
        ctx->h_stage[ctx->slot] = nullptr;
        ctx->h_stage_cap[ctx->slot] = 0;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_stage[ctx->slot], need));
        ctx->h_stage_cap[ctx->slot] = need;
        memset(ctx->h_stage[ctx->slot], 0, 128);        // (totals and the completion word of a fresh block: never a stale sequence number)
    }
    return DRT_OK;
}

// asynchronous host-buffer renders: gradients and totals of the frame -> the pinned block, written by the device (one
// small launch in stream order; the image got there from the finishing kernels)
// The LAST launch of a synchronous host-buffer render: with everything of the frame stored (this wave's own words fenced
// behind it), it sets the block's completion word to `seq` -- the caller polls that word instead of waiting for the
// runtime to notice the end of the stream (render_finish).
__global__ void __launch_bounds__(DRT_WAVE) k_results_to_host(const double* __restrict__ grad, int n_grad, const uint8_t* __restrict__ requires_grad_dev,
                                                              const unsigned long long* __restrict__ totals, double* __restrict__ h_grad,
                                                              unsigned long long* __restrict__ h_totals, unsigned long long* __restrict__ h_done = nullptr,
                                                              unsigned long long seq = 0)
{
    (void)requires_grad_dev;
    for (int i = threadIdx.x; i < n_grad; i += DRT_WAVE)
        h_grad[i] = grad[i];
    if (totals && threadIdx.x < DRT_TOTAL_WORDS)
        h_totals[threadIdx.x] = totals[threadIdx.x];
    if (h_done) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(h_done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// the same for the two-stream form, image included: the rows of this shard (full-frame layout on both sides), float by float
// or, unsharded, 16 bytes per lane; a few blocks next to the following frame's kernels -- the PCIe link is the limit, not the CUs
__global__ void __launch_bounds__(DRT_BLOCK) k_frame_to_host(const float* __restrict__ img, float* __restrict__ h_img, uint32_t row_floats,
                                                            uint32_t n_local_rows, uint32_t band, uint32_t n_shards, uint32_t shard,
                                                            const double* __restrict__ grad, int n_grad,
                                                            const unsigned long long* __restrict__ totals, double* __restrict__ h_grad,
                                                            unsigned long long* __restrict__ h_totals)
{
    const uint64_t n = img ? (uint64_t)n_local_rows * row_floats : 0;
    const uint64_t stride = (uint64_t)gridDim.x * DRT_BLOCK, first = (uint64_t)blockIdx.x * DRT_BLOCK + threadIdx.x;
    if (n_shards <= 1 && (n & 3u) == 0) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(img);
        float4* __restrict__ dst = reinterpret_cast<float4*>(h_img);
        for (uint64_t i = first; i < n / 4; i += stride)
            dst[i] = src[i];
    } else {
        for (uint64_t i = first; i < n; i += stride) {
            const uint32_t lr = (uint32_t)(i / row_floats), c = (uint32_t)(i - (uint64_t)lr * row_floats);
            uint32_t y = lr;
            if (n_shards > 1) {
                const uint32_t b = lr / band, r = lr - b * band;
                y = (b * n_shards + shard) * band + r;
            }
            const size_t at = (size_t)y * row_floats + c;
            h_img[at] = img[at];
        }
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < n_grad; i += DRT_BLOCK)
            h_grad[i] = grad[i];
        if (threadIdx.x < DRT_TOTAL_WORDS)
            h_totals[threadIdx.x] = totals[threadIdx.x];
    }
}

// phase 3: results on their way to the caller (device pointers: a copy on the stream; host buffers: DMA into the
// context's pinned staging block -- only the rows of this shard)
static int render_collect(drt_hip_ctx* ctx, bool with_grad = true, hipStream_t cs = nullptr)
{
    RenderJob& j = ctx->job;
    if (!cs)
        cs = ctx->stream;                  // (an asynchronous render copies on the context's copy stream)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t npix_all = (size_t)j.cam.width * j.cam.height;
    // parameters that do not require grad keep a zero gradient (vector.hpp:156-162)
    if (j.backward && j.dev_out && j.out_param_grad && with_grad) {
        for (int p = 0; p < ctx->n_user_params; ++p)
            if (!ctx->requires_grad[p])
                HIPCHK(ctx, hipMemsetAsync((double*)ctx->grad[ctx->slot].p + (size_t)p * 3, 0, 3 * sizeof(double), cs));
        HIPCHK(ctx, hipMemcpyAsync(j.out_param_grad, ctx->grad[ctx->slot].p, (size_t)ctx->n_user_params * 3 * sizeof(double), hipMemcpyDeviceToDevice, cs));
    }
    j.sync = !j.dev_out || (j.rp.flags & DRT_RENDER_SYNC) || j.timing || j.stats;
    {
        const int rc = ensure_stage(ctx, j);
        if (rc != DRT_OK) return rc;
    }
    (void)npix_all;
    if (j.copy_kernel) {
        // (asynchronous host-buffer render, two-stream form: everything of the frame crosses the link in one launch on the
        //  copy stream while the next frame's kernels run)
        ctx->h_segments = 0;
        j.want_segments = j.stats && j.n_count_words;
        const bool img = j.out_rgb && j.n_local_pixels && !j.direct_out;   // (a pinned caller buffer holds its image already)
        const int copy_blocks = tuning().copy_blocks;
        hipLaunchKernelGGL(k_frame_to_host, dim3(copy_blocks), dim3(DRT_BLOCK), 0, cs, img ? (const float*)j.d_out : (const float*)nullptr,
                           (float*)(ctx->h_stage[ctx->slot] + j.off_img), (uint32_t)j.cam.width * 3u,
                           (uint32_t)(j.n_local_pixels / (uint32_t)j.cam.width), (uint32_t)j.band, (uint32_t)j.n_shards, (uint32_t)j.shard,
                           (const double*)ctx->grad[ctx->slot].p, (j.backward && j.out_param_grad && with_grad) ? ctx->n_user_params * 3 : 0,
                           (const unsigned long long*)ctx->segtotal[ctx->slot].p, (double*)(ctx->h_stage[ctx->slot] + j.off_grad),
                           (unsigned long long*)ctx->h_stage[ctx->slot]);
        HIPCHK(ctx, hipGetLastError());
        return DRT_OK;
    }
    ctx->h_segments = 0;
    j.want_segments = j.stats && j.n_count_words;
    if (!j.dev_out) {
        const size_t row_bytes = (size_t)j.cam.width * 3 * sizeof(float);
        hipError_t e = hipSuccess;
        auto rows_to_stage = [&](const float* d_src, size_t off) {
            for_each_band(j.cam.height, j.band, j.n_shards, j.shard, [&](int y0, int y1) {
                if (e == hipSuccess)
                    e = hipMemcpyAsync(ctx->h_stage[ctx->slot] + off + (size_t)y0 * row_bytes, (const uint8_t*)d_src + (size_t)y0 * row_bytes,
                                       (size_t)(y1 - y0) * row_bytes, hipMemcpyDeviceToHost, cs);
            });
        };
        // (the image is where it belongs already when the finishing kernels stored it into the pinned block -- synchronous
        //  renders, the one-stream asynchronous form -- or into the caller's own pinned buffer)
        if (j.out_rgb && j.n_local_pixels && !j.zero_copy && !j.direct_out)
            rows_to_stage(j.d_out, j.off_img);
        if (j.gimg_param >= 0 && j.out_gimg && j.n_local_pixels && !j.direct_gimg)
            rows_to_stage(j.d_gimg, j.off_gimg);
        HIPCHK(ctx, e);
        // gradients, totals and the completion word: ONE small launch behind everything else of the frame
        j.done_seq = ++ctx->done_seq;
        hipLaunchKernelGGL(k_results_to_host, dim3(1), dim3(DRT_WAVE), 0, cs, (const double*)ctx->grad[ctx->slot].p,
                           (j.backward && j.out_param_grad && with_grad) ? ctx->n_user_params * 3 : 0, (const uint8_t*)nullptr,
                           j.want_segments ? (const unsigned long long*)ctx->segtotal[ctx->slot].p : (const unsigned long long*)nullptr,
                           (double*)(ctx->h_stage[ctx->slot] + j.off_grad), (unsigned long long*)ctx->h_stage[ctx->slot],
                           (unsigned long long*)(ctx->h_stage[ctx->slot] + DRT_STAGE_DONE), (unsigned long long)j.done_seq);
        HIPCHK(ctx, hipGetLastError());
    } else if (j.want_segments) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage[ctx->slot], ctx->segtotal[ctx->slot].p, DRT_TOTAL_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
    }
    return DRT_OK;
}

// phase 4: wait (unless the caller asked for an asynchronous device-pointer render), hand over, statistics
static int render_finish(drt_hip_ctx* ctx, bool with_grad = true, hipEvent_t done = nullptr)
{
    RenderJob& j = ctx->job;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (done)
        HIPCHK(ctx, hipEventSynchronize(done));       // (an asynchronous render: its copies are complete; later frames may still run)
    else if (j.sync) {
        // A host-buffer render ends with a launch that sets the block's completion word: poll it (the runtime's own wait
        // notices the end of a stream ~10-20 us late, a third of a small frame) -- for a bounded time, so that a fault on the
        // device still surfaces through hipStreamSynchronize.
        bool seen = false;
        if (j.done_seq && !j.timing && tuning().sync_spin_us > 0 && ctx->h_stage[ctx->slot]) {
            const volatile unsigned long long* w = (const volatile unsigned long long*)(ctx->h_stage[ctx->slot] + DRT_STAGE_DONE);
            const auto t_spin = std::chrono::steady_clock::now();
            for (uint32_t it = 0;; ++it) {
                if (__atomic_load_n(w, __ATOMIC_ACQUIRE) == (unsigned long long)j.done_seq) { seen = true; break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#elif defined(__aarch64__)
                __asm__ __volatile__("isb" ::: "memory");
#else
                std::this_thread::yield();
#endif
                if ((it & 255u) == 255u &&
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_spin).count() > (double)tuning().sync_spin_us)
                    break;
            }
        }
        if (!seen)
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        else {
            // (the word was written by the frame's LAST launch, so the stream is through -- a query costs a microsecond and
            //  surfaces a fault of this frame here instead of at some later call)
            const hipError_t eq = hipStreamQuery(ctx->stream);
            if (eq != hipSuccess && eq != hipErrorNotReady) {
                ctx->err = std::string("render: ") + hipGetErrorString(eq);
                return DRT_ERR_HIP;
            }
        }
    }
    unsigned long long h_tot[DRT_TOTAL_WORDS] = {0};
    if (j.want_segments)
        memcpy(h_tot, ctx->h_stage[ctx->slot], sizeof h_tot);
    ctx->h_segments = h_tot[0];
    if (!j.dev_out) {
        const size_t row_bytes = (size_t)j.cam.width * 3 * sizeof(float);
        auto rows_to_caller = [&](float* dst, size_t off) {
            for_each_band(j.cam.height, j.band, j.n_shards, j.shard, [&](int y0, int y1) {
                memcpy((uint8_t*)dst + (size_t)y0 * row_bytes, ctx->h_stage[ctx->slot] + off + (size_t)y0 * row_bytes, (size_t)(y1 - y0) * row_bytes);
            });
        };
        if (j.out_rgb && j.n_local_pixels && !j.direct_out)
            rows_to_caller(j.out_rgb, j.off_img);
        if (j.gimg_param >= 0 && j.out_gimg && j.n_local_pixels && !j.direct_gimg)
            rows_to_caller(j.out_gimg, j.off_gimg);
        if (j.backward && j.out_param_grad && with_grad) {
            memcpy(j.out_param_grad, ctx->h_stage[ctx->slot] + j.off_grad, j.grad_bytes);
            for (int p = 0; p < ctx->n_user_params; ++p)
                if (!ctx->requires_grad[p])
                    j.out_param_grad[p * 3] = j.out_param_grad[p * 3 + 1] = j.out_param_grad[p * 3 + 2] = 0.0;
        }
    }
    if (j.stats) {
        drt_hip_stats& st = j.st;
        st.segments = h_tot[0];
        st.queue_rays_read = h_tot[1];
        st.queue_rays_written = h_tot[2];
        st.capped_paths = h_tot[3];
        st.bvh_bytes = ctx->has_mesh ? ctx->bvh_bytes : 0;
        st.jit_ms = ctx->jit_ms;
        st.units[DRT_K_INTERSECT] = h_tot[4];              // rays k_intersect tested (mesh scenes: the camera rays only)
        st.units[DRT_K_INTERSECT_MESH] = h_tot[5];         // candidate rays the BVH walk took (those that reach the mesh bounds)
        st.units[DRT_K_SHADE] = st.launches[DRT_K_SHADE] ? st.segments : 0;
        st.units[DRT_K_PATH] = st.launches[DRT_K_PATH] ? st.segments : 0;
        st.units[DRT_K_BACKWARD] = j.backward && st.launches[DRT_K_BACKWARD] ? st.segments : 0;
        if (j.timing) {
            for (const TimedLaunch& t : ctx->timed) {
                float ms = 0;
                HIPCHK(ctx, hipEventElapsedTime(&ms, t.e0, t.e1));
                st.ms_kernel[t.kernel] += (double)ms;
            }
        }
        st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - j.t0).count();
        *j.stats = st;
    }
    return DRT_OK;
}

__global__ void __launch_bounds__(DRT_BLOCK) k_add_f64(double* __restrict__ dst, const double* __restrict__ src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] += src[i];
}

// A group context: every phase on ALL members before the next one, so n devices run concurrently under one call
// (the launch phase from one host thread per member, the others from the caller's).  The gradient: members that share a device are added to their leader on that device (stream-ordered
// through events), then ONE ncclAllReduce over the leaders -- the single collective of the path.
static int render_group(drt_hip_ctx* g, const drt_camera_desc* cam, const drt_render_params* rp, const float* adjoint_rgb,
                        float* out_rgb, double* out_param_grad, drt_hip_stats* stats, int gimg_param, float* out_gimg)
{
    if (!cam || !rp)
        return fail(g, DRT_ERR_INVALID, "render: bad camera or render parameters");
    if (rp->flags & DRT_RENDER_DEVICE_OUT)
        return fail(g, DRT_ERR_UNSUPPORTED, "render: a group context returns through host buffers (no DRT_RENDER_DEVICE_OUT)");
    const bool backward = (rp->flags & DRT_RENDER_BACKWARD) != 0;
    if (backward && !out_param_grad && gimg_param < 0)
        return fail(g, DRT_ERR_INVALID, "render: DRT_RENDER_BACKWARD needs out_param_grad");
    const auto t0 = std::chrono::steady_clock::now();
    const int n = (int)g->members.size();
    const int outer = rp->n_shards > 1 ? rp->n_shards : 1, outer_shard = rp->n_shards > 1 ? rp->shard : 0;
    if (outer_shard < 0 || outer_shard >= outer)
        return fail(g, DRT_ERR_INVALID, "render: shard out of range");
    std::vector<drt_hip_stats> mstats((size_t)n);
    auto member_fail = [&](int i, int rc) { g->err = "device " + std::to_string(g->members[i]->device) + ": " + g->members[i]->err; return rc; };
    int rc;
    {
        // The launch phase can block the host -- a pageable adjoint image is copied synchronously, and deep roulette-terminated
        // renders on the queue route ask the device every few bounces whether any path is still alive -- so every member
        // enqueues its share from its own host thread: the devices start together whatever one member's launch waits for.
        const bool threads_env = tuning().group_threads;
        std::vector<int> rcs((size_t)n, DRT_OK);
        auto launch_member = [&](int i) {
            drt_render_params r = *rp;
            r.n_shards = outer * n;
            r.shard = outer_shard * n + i;
            r.flags &= ~(uint32_t)(DRT_RENDER_ALLREDUCE | DRT_RENDER_ALLREDUCE_ASYNC);    // the group reduces below
            rcs[(size_t)i] = render_launch(g->members[i], cam, &r, adjoint_rgb, out_rgb, out_param_grad,
                                           stats ? &mstats[i] : nullptr, gimg_param, out_gimg);
        };
        if (threads_env && n > 1) {
            std::vector<std::thread> workers;
            for (int i = 1; i < n; ++i)
                workers.emplace_back(launch_member, i);
            launch_member(0);
            for (std::thread& w : workers)
                w.join();
        } else {
            for (int i = 0; i < n; ++i)
                launch_member(i);
        }
        for (int i = 0; i < n; ++i)
            if (rcs[(size_t)i] != DRT_OK)
                return member_fail(i, rcs[(size_t)i]);
    }
    if (backward && gimg_param < 0) {
        const int words = g->members[0]->n_user_params * 3;
        for (int i = 0; i < n; ++i) {
            drt_hip_ctx* m = g->members[i];
            if (g->leader[i] == i)
                continue;
            drt_hip_ctx* l = g->members[g->leader[i]];
            HIPCHK(m, hipSetDevice(m->device));
            HIPCHK(m, hipEventRecord(m->ev_done, m->stream));
            HIPCHK(l, hipStreamWaitEvent(l->stream, m->ev_done, 0));
            hipLaunchKernelGGL(k_add_f64, dim3((words + DRT_BLOCK - 1) / DRT_BLOCK), dim3(DRT_BLOCK), 0, l->stream,
                               (double*)l->grad[l->slot].p, (const double*)m->grad[m->slot].p, words);
        }
        ncclResult_t r = ncclGroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i) {
            drt_hip_ctx* m = g->members[i];
            if (g->leader[i] != i)
                continue;
            r = ncclAllReduce(m->grad[m->slot].p, m->grad[m->slot].p, (size_t)words, ncclDouble, ncclSum, m->comm, m->stream);
        }
        const ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess) {
            g->err = std::string("ncclAllReduce (group): ") + ncclGetErrorString(r != ncclSuccess ? r : r2);
            return DRT_ERR_COMM;
        }
    }
    for (int i = 0; i < n; ++i)        // every member copies its rows; member 0 (a leader) the reduced gradient
        if ((rc = render_collect(g->members[i], i == 0)) != DRT_OK)
            return member_fail(i, rc);
    for (int i = 0; i < n; ++i)
        if ((rc = render_finish(g->members[i], i == 0)) != DRT_OK)
            return member_fail(i, rc);
    if (stats) {
        drt_hip_stats st = mstats[0];
        for (int i = 1; i < n; ++i) {
            st.paths += mstats[i].paths;
            st.segments += mstats[i].segments;
            st.batches += mstats[i].batches;
            st.queue_rays_read += mstats[i].queue_rays_read;
            st.queue_rays_written += mstats[i].queue_rays_written;
            st.capped_paths += mstats[i].capped_paths;
            st.path_bytes += mstats[i].path_bytes;
            st.jit_ms += mstats[i].jit_ms;
            for (int k = 0; k < DRT_K_COUNT; ++k) {
                st.units[k] += mstats[i].units[k];
                if (mstats[i].ms_kernel[k] > st.ms_kernel[k])
                    st.ms_kernel[k] = mstats[i].ms_kernel[k];      // devices run side by side: the slowest counts
            }
        }
        st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        *stats = st;
    }
    return DRT_OK;
}

static int render_common(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                         const float* adjoint_rgb, float* out_rgb, double* out_param_grad, drt_hip_stats* stats,
                         int gimg_param, float* out_gimg)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!ctx->members.empty())
        return render_group(ctx, cam, rp, adjoint_rgb, out_rgb, out_param_grad, stats, gimg_param, out_gimg);
    for (int i = 0; i < DRT_HIP_FRAMES_IN_FLIGHT; ++i)
        if (ctx->in_flight[i])
            return fail(ctx, DRT_ERR_INVALID, "render: asynchronous frames are in flight -- drt_hip_wait for them first");
    int rc;
    // DRT_RENDER_ALLREDUCE_ASYNC (device buffers, a communicator): the all-reduce and the copy of the reduced gradient run on
    // the context's second stream while the NEXT render's kernels run on the first; the two gradient sets alternate, and a
    // render only waits for the all-reduce of the render before the previous one (long finished)
    const bool ar_async = rp && gimg_param < 0 && (rp->flags & DRT_RENDER_ALLREDUCE_ASYNC) && (rp->flags & DRT_RENDER_BACKWARD) &&
                          (rp->flags & DRT_RENDER_DEVICE_OUT) && ctx->comm;
    // Device-pointer renders that do not wait (no DRT_RENDER_SYNC, no statistics): consecutive frames alternate between the
    // context's two sets of per-frame buffers, so that their k_path grids can overlap (render_impl: path_stream)
    const bool dev_async = rp && gimg_param < 0 && (rp->flags & DRT_RENDER_DEVICE_OUT) &&
                           !(rp->flags & (DRT_RENDER_SYNC | DRT_RENDER_TIMING)) && !stats;
    if (ar_async || dev_async) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if ((rc = ensure_copy_stream(ctx)) != DRT_OK) return rc;
        ctx->slot = (int)(ctx->dev_frames & 1);
        if (ar_async && ctx->slot_used[ctx->slot])
            HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copied[ctx->slot], 0));
    }
    ctx->overlap_next = (ar_async || dev_async) && !(rp->flags & DRT_RENDER_SERIAL);
    // (host buffers: the finishing kernels store the image into the context's pinned block -- no copy launch behind them)
    ctx->zero_copy_next = rp && !(rp->flags & DRT_RENDER_DEVICE_OUT) && tuning().sync_zero_copy;
    rc = render_launch(ctx, cam, rp, adjoint_rgb, out_rgb, out_param_grad, stats, gimg_param, out_gimg);
    ctx->overlap_next = false;
    ctx->zero_copy_next = false;
    if (rc != DRT_OK) {
        abort_comm_after_failure(ctx, rp);
        ctx->slot = 0;
        return rc;
    }
    if (ar_async) {
        const int slot = ctx->slot;
        hipError_t e = hipEventRecord(ctx->ev_rendered[slot], ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_rendered[slot], 0);
        if (e != hipSuccess) { ctx->err = std::string("render: ") + hipGetErrorString(e); ctx->slot = 0; return DRT_ERR_HIP; }
        if ((rc = render_reduce(ctx, ctx->copy_stream)) == DRT_OK) rc = render_collect(ctx, true, ctx->copy_stream);
        if (rc == DRT_OK && hipEventRecord(ctx->ev_copied[slot], ctx->copy_stream) != hipSuccess) rc = DRT_ERR_HIP;
        if (rc == DRT_OK) ctx->slot_used[slot] = true;
        ++ctx->dev_frames;
        if (rc == DRT_OK && ctx->job.sync)
            HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (rc == DRT_OK) rc = render_finish(ctx);
        ctx->slot = 0;
        return rc;
    }
    if ((rc = render_reduce(ctx)) == DRT_OK) rc = render_collect(ctx);
    if (rc == DRT_OK && dev_async) {            // the slot's buffers are free once the stream has come this far
        if (hipEventRecord(ctx->ev_copied[ctx->slot], ctx->stream) != hipSuccess) rc = DRT_ERR_HIP;
        else ctx->slot_used[ctx->slot] = true;
        ++ctx->dev_frames;
    }
    if (rc == DRT_OK) rc = render_finish(ctx);
    ctx->slot = 0;
    return rc;
}

