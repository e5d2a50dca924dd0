// drt_hip.hip -- libdrt_hip.so: the entry points of the C ABI (include/drt_hip.h).  The one translation unit of the library;
// the host runtime behind the entry points is in headers by topic:
//   drt_ctx.h          the context (device, streams, buffers), the state of a render call, helpers
//   drt_tuning.h       every DRT_HIP_* environment variable, read once (listed in INTEGRATION.md section 2)
//   drt_scene.h        upload_scene / update_params: POD scene -> device records, BVH
//   drt_jit.h          k_path compiled for a scene's shape kinds at run time (hiprtc)
//   drt_render_impl.h  one shard's render enqueued: the k_path route and the queue wavefront
//   drt_render.h       a render call in phases; group contexts; asynchronous frames; the all-reduce
// and the kernels in headers by topic too:
//   drt_kernels.h      K1-K5 of the queue wavefront + what all kernels share (RNG, camera, analytic closest hit, BxDF sampler)
//   drt_walk.h         K2 on triangles: the BVH walk
//   drt_backward.h     K6 / K7: the tape's reverse sweep, gradient accumulators, the fixed-order reduction
//   drt_chain.h        the unbiased operator's adjoint rounds on the queue wavefront
//   drt_path.h         k_path / k_path_unbiased: the whole path in one launch (analytic scenes)
//   drt_path_mesh.h    k_path_mesh: the same with the BVH walk inside (small frames of mesh scenes)
#include "drt_kernels.h"
#include "drt_walk.h"
#include "drt_backward.h"
#include "drt_chain.h"
#include "drt_path.h"
#include "drt_path_mesh.h"
#include "drt_bvh.h"
#include "drt_jit.h"

#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>


#include "drt_tuning.h"
#include "drt_ctx.h"
#include "drt_scene.h"
#include "drt_render_impl.h"
#include "drt_render.h"

extern "C" {

int drt_hip_abi_version(void) { return DRT_HIP_ABI_VERSION; }

int drt_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int drt_hip_create(int device_id, drt_hip_ctx** out)
{
    if (!out)
        return DRT_ERR_INVALID;
    *out = nullptr;
    int n = drt_hip_device_count();
    if (n <= 0 || device_id < 0 || device_id >= n)
        return DRT_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess)
        return DRT_ERR_NO_DEVICE;
    drt_hip_ctx* ctx = new drt_hip_ctx();
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) {
        ctx->n_cu = prop.multiProcessorCount;
        ctx->device_mem = (uint64_t)prop.totalGlobalMem;
        if (prop.gcnArchName[0])
            ctx->arch = prop.gcnArchName;
    }
    ctx->jit_mode = tuning().jit;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return DRT_ERR_HIP;
    }
    {   // Every stream of the context is made HERE, in this order, before the process makes any other: which of them run
        // side by side depends on the order HIP has seen them in (measured: the two k_path streams made later, next to the
        // copy stream, never overlapped their grids; made here they do -- and the copy stream made later, after them, no
        // longer overlapped its launch with the next frame: 0.81 -> 0.90 ms through host buffers).
        for (int i = 0; i < 2 && tuning().overlap_frames; ++i)
            if (hipStreamCreateWithFlags(&ctx->path_stream[i], hipStreamNonBlocking) != hipSuccess)
                ctx->path_stream[i] = nullptr;
        if (!ctx->path_stream[1]) ctx->path_stream[0] = nullptr;
        // (highest priority: the copies and the all-reduce of frame i must not queue behind the kernels of frame i + 1)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, greatest) != hipSuccess)
            ctx->copy_stream = nullptr;
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&ctx->ev_lane_free[i], hipEventDisableTiming) != hipSuccess)
                ctx->ev_lane_free[i] = nullptr;
        if (hipEventCreateWithFlags(&ctx->ev_params, hipEventDisableTiming) != hipSuccess)
            ctx->ev_params = nullptr;
    }
    {   // the BVH walk is a persistent kernel whose waves own strided streams of rays: its grid must be exactly what
        // is resident at once (more blocks would run as a second round behind the first, at half the occupancy)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_intersect_mesh<float>, DRT_BLOCK, 0) == hipSuccess && nb > 0)
            ctx->mesh_blocks_per_cu = nb;
        (void)hipGetLastError();
        if (tuning().mesh_blocks_per_cu > 0)
            ctx->mesh_blocks_per_cu = tuning().mesh_blocks_per_cu;
    }
    *out = ctx;
    return DRT_OK;
}

void drt_hip_destroy(drt_hip_ctx* ctx)
{
    if (!ctx)
        return;
    if (!ctx->members.empty() || !ctx->stream) {          // a group context owns members, nothing else
        for (drt_hip_ctx* m : ctx->members)
            drt_hip_destroy(m);
        delete ctx;
        return;
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream)
        (void)hipStreamSynchronize(ctx->copy_stream);
    if (ctx->comm)
        (void)ncclCommDestroy(ctx->comm);
    if (ctx->ev_done)
        (void)hipEventDestroy(ctx->ev_done);
    DevBuf* bufs[] = {&ctx->hist_ovf[0], &ctx->hist_ovf[1], &ctx->mesh_ovf[0], &ctx->mesh_ovf[1], &ctx->fpart2, &ctx->gpart2, &ctx->counts2, &ctx->fpart, &ctx->gpix, &ctx->cand[0], &ctx->cand[1], &ctx->cand_a[0], &ctx->cand_a[1], &ctx->cand_b[0], &ctx->cand_b[1], &ctx->cand_count[0], &ctx->cand_count[1], &ctx->ray_a[0], &ctx->ray_a[1], &ctx->ray_a[2], &ctx->ray_b[0], &ctx->ray_b[1], &ctx->ray_b[2], &ctx->ray_id[0], &ctx->ray_id[1], &ctx->ray_id[2], &ctx->hit, &ctx->hit2, &ctx->hit3, &ctx->lacc, &ctx->gpath, &ctx->gfilm, &ctx->gimg_out, &ctx->tape, &ctx->nv,
                      &ctx->ch_cva, &ctx->ch_cvb, &ctx->ch_cvh, &ctx->ch_nxa, &ctx->ch_nxb, &ctx->ch_nxh, &ctx->ch_g,
                      &ctx->ch_w, &ctx->ch_ids, &ctx->ch_ndraw, &ctx->ch_dbase, &ctx->counts, &ctx->film, &ctx->gpart, &ctx->adjoint};
    for (DevBuf* b : bufs)
        release(*b);
    for (int i = 0; i < DRT_HIP_FRAMES_IN_FLIGHT; ++i) {
        release(ctx->segtotal[i]);
        release(ctx->grad[i]);
        release(ctx->out[i]);
    }
    release_mesh(ctx);
    for (hipModule_t m : ctx->jit_modules)
        (void)hipModuleUnload(m);
    drt_jit::wait_idle();                   // (a compile this context started in the background is allowed to finish)
    if (ctx->d_scene_f) (void)hipFree(ctx->d_scene_f);
    if (ctx->d_scene_d) (void)hipFree(ctx->d_scene_d);
    if (ctx->d_params_f) (void)hipFree(ctx->d_params_f);
    if (ctx->d_params_d) (void)hipFree(ctx->d_params_d);
    if (ctx->h_probe)
        (void)hipHostFree(ctx->h_probe);
    if (ctx->h_params)
        (void)hipHostFree(ctx->h_params);
    for (const drt_hip_ctx::PinnedRange& r : ctx->pinned)
        (void)hipHostUnregister(r.host);
    for (int i = 0; i < DRT_HIP_FRAMES_IN_FLIGHT; ++i) {
        if (ctx->h_stage[i])
            (void)hipHostFree(ctx->h_stage[i]);
        if (ctx->ev_rendered[i]) (void)hipEventDestroy(ctx->ev_rendered[i]);
        if (ctx->ev_copied[i]) (void)hipEventDestroy(ctx->ev_copied[i]);
    }
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    for (int i = 0; i < 2; ++i) {
        if (ctx->path_stream[i]) { (void)hipStreamSynchronize(ctx->path_stream[i]); (void)hipStreamDestroy(ctx->path_stream[i]); }
        if (ctx->ev_begin[i]) (void)hipEventDestroy(ctx->ev_begin[i]);
        if (ctx->ev_path[i]) (void)hipEventDestroy(ctx->ev_path[i]);
        if (ctx->ev_lane_free[i]) (void)hipEventDestroy(ctx->ev_lane_free[i]);
    }
    if (ctx->ev_params) (void)hipEventDestroy(ctx->ev_params);
    release(ctx->probe);
    for (hipEvent_t e : ctx->event_pool)
        (void)hipEventDestroy(e);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

} // extern "C"

extern "C" {

// ---- caller buffers the finishing kernels write directly (ABI v7) -----------------------------------------------------
int drt_hip_pin_host(drt_hip_ctx* ctx, void* ptr, size_t bytes)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!ctx->members.empty())
        return fail(ctx, DRT_ERR_UNSUPPORTED, "pin_host: not on a group context");
    if (!ptr || bytes == 0)
        return fail(ctx, DRT_ERR_INVALID, "pin_host: NULL or empty range");
    for (const drt_hip_ctx::PinnedRange& r : ctx->pinned)
        if ((uint8_t*)ptr < r.host + r.bytes && r.host < (uint8_t*)ptr + bytes)
            return fail(ctx, DRT_ERR_INVALID, "pin_host: the range overlaps one that is pinned already");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    void* dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess || !dev) {
        (void)hipHostUnregister(ptr);
        (void)hipGetLastError();
        return fail(ctx, DRT_ERR_HIP, "pin_host: the range cannot be mapped into the device's address space");
    }
    ctx->pinned.push_back({(uint8_t*)ptr, bytes, (uint8_t*)dev});
    return DRT_OK;
}

int drt_hip_unpin_host(drt_hip_ctx* ctx, void* ptr)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    for (size_t i = 0; i < ctx->pinned.size(); ++i)
        if (ctx->pinned[i].host == (uint8_t*)ptr) {
            for (int f = 0; f < DRT_HIP_FRAMES_IN_FLIGHT; ++f)
                if (ctx->in_flight[f])
                    return fail(ctx, DRT_ERR_INVALID, "unpin_host: asynchronous frames are in flight -- drt_hip_wait for them first");
            HIPCHK(ctx, hipSetDevice(ctx->device));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, hipHostUnregister(ptr));
            ctx->pinned.erase(ctx->pinned.begin() + (long)i);
            return DRT_OK;
        }
    return fail(ctx, DRT_ERR_INVALID, "unpin_host: not the start of a pinned range");
}

int drt_hip_render(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                   const float* adjoint_rgb, float* out_rgb, double* out_param_grad, drt_hip_stats* stats)
{
    return render_common(ctx, cam, rp, adjoint_rgb, out_rgb, out_param_grad, stats, -1, nullptr);
}

int drt_hip_render_gradient_image(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                                  int32_t param, const float* adjoint_rgb, float* out_rgb, float* out_grad_rgb,
                                  drt_hip_stats* stats)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    drt_hip_ctx* first = ctx->members.empty() ? ctx : ctx->members[0];
    if (!first->has_scene)
        return fail(ctx, DRT_ERR_NO_SCENE, "render before upload_scene");
    if (!rp || !out_grad_rgb || param < 0 || param >= first->n_user_params)
        return fail(ctx, DRT_ERR_INVALID, "gradient image: bad parameter index or NULL output");
    drt_render_params r = *rp;
    r.flags |= DRT_RENDER_BACKWARD;
    return render_common(ctx, cam, &r, adjoint_rgb, out_rgb, nullptr, stats, param, out_grad_rgb);
}

// ---- asynchronous host-buffer renders ---------------------------------------------------------------
// drt_hip_render returns when the results are in the caller's buffers: every frame pays a 3 MB device-to-host copy and a
// stream synchronisation with the GPU idle meanwhile (config 3: 1.13 instead of 0.86 ms per frame).  An optimisation loop
// that renders frame after frame (render.cpp:72-90 inside a gradient-descent loop) overlaps them: frame i's results travel to
// a pinned block on a second stream while frame i + 1's kernels run; drt_hip_wait hands them over.
int drt_hip_render_async(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp, const float* adjoint_rgb,
                         float* out_rgb, double* out_param_grad, uint64_t* ticket)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!ticket)
        return fail(ctx, DRT_ERR_INVALID, "render_async: ticket is NULL");
    *ticket = 0;
    if (!ctx->members.empty())
        return fail(ctx, DRT_ERR_UNSUPPORTED, "render_async: not on a group context");
    if (rp && (rp->flags & (DRT_RENDER_DEVICE_OUT | DRT_RENDER_TIMING)))
        return fail(ctx, DRT_ERR_INVALID, "render_async: host buffers only, no per-kernel timing (use drt_hip_render)");
    const uint64_t t = ctx->next_ticket;
    const int slot = (int)(t % DRT_HIP_FRAMES_IN_FLIGHT);
    if (ctx->in_flight[slot])
        return fail(ctx, DRT_ERR_INVALID, "render_async: four frames are in flight already -- drt_hip_wait for the oldest one first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    {
        const int rc0 = ensure_copy_stream(ctx);
        if (rc0 != DRT_OK) return rc0;
    }
    ctx->slot = slot;
    static drt_hip_stats sink;            // (render_launch only notes that totals are wanted; drt_hip_wait fills the caller's)
    // The frame's kernels run on the context's first stream and leave image, gradients and totals in the device buffers of the
    // frame's slot; ONE copy launch on the second stream (k_frame_to_host: zero-copy stores into the pinned block -- no DMA
    // engine, no hipMemcpy) carries them across the link while the next frame's kernels run on the first; an event marks the
    // end.  Config 3, per frame: 0.84-0.87 ms against 0.834 on device pointers and 1.10 for the synchronous call.
    // (Measured before it, round 3: the same copy as hipMemcpyAsync on the second stream -- 0.85 ms under ROCm 7.2's runtime
    // but 1.3 ms, slower than the synchronous call, when the process had loaded PyTorch's bundled ROCm 7.0 runtime first, as
    // bench.py does; and everything on ONE stream, the finishing kernels storing the image straight into the pinned block
    // (DRT_HIP_ASYNC_COPY=inline, still there): 0.89-0.92 ms -- the 3 MB cross the link inside the frame's critical path.)
    const bool two_streams = !tuning().async_copy_inline;
    ctx->zero_copy_next = !two_streams;
    // (the k_path grids of consecutive frames overlap -- render_impl: path_stream --: frame t shares its stream and its set of
    //  partial sums with frame t - 2, whose finishing launch, on the context's stream, must have read them)
    ctx->overlap_next = two_streams && !(rp && (rp->flags & DRT_RENDER_SERIAL));
    ctx->stage_adjoint_next = true;
    int rc = render_launch(ctx, cam, rp, adjoint_rgb, out_rgb, out_param_grad, &sink, -1, nullptr);
    ctx->stage_adjoint_next = false;
    ctx->zero_copy_next = false;
    ctx->overlap_next = false;
    if (rc != DRT_OK)
        abort_comm_after_failure(ctx, rp);
    if (rc == DRT_OK) rc = render_reduce(ctx);
    hipStream_t done_on = ctx->stream;
    if (rc == DRT_OK && two_streams) {
        ctx->job.copy_kernel = true;
        hipError_t e = hipEventRecord(ctx->ev_rendered[slot], ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, ctx->ev_rendered[slot], 0);
        if (e != hipSuccess) { ctx->err = std::string("render_async: ") + hipGetErrorString(e); rc = DRT_ERR_HIP; }
        done_on = ctx->copy_stream;
    }
    if (rc == DRT_OK) rc = render_collect(ctx, true, done_on);
    if (rc == DRT_OK && hipEventRecord(ctx->ev_copied[slot], done_on) != hipSuccess) {
        ctx->err = "render_async: hipEventRecord failed";
        rc = DRT_ERR_HIP;
    }
    if (rc != DRT_OK) {                   // nothing of this frame stays in flight
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->copy_stream);
        ctx->slot = 0;
        return rc;
    }
    ctx->slot_used[slot] = true;
    ctx->pending[slot] = ctx->job;
    ctx->in_flight[slot] = true;
    ctx->next_ticket = t + 1;
    ctx->slot = 0;
    *ticket = t;
    return DRT_OK;
}

int drt_hip_wait(drt_hip_ctx* ctx, uint64_t ticket, drt_hip_stats* stats)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    const int slot = (int)(ticket % DRT_HIP_FRAMES_IN_FLIGHT);
    if (ticket == 0 || ticket >= ctx->next_ticket || !ctx->in_flight[slot] || ticket + DRT_HIP_FRAMES_IN_FLIGHT < ctx->next_ticket)
        return fail(ctx, DRT_ERR_INVALID, "wait: no such frame in flight");
    ctx->slot = slot;
    ctx->job = ctx->pending[slot];
    ctx->job.stats = stats;               // (NULL: no statistics wanted)
    const int rc = render_finish(ctx, true, ctx->ev_copied[slot]);
    ctx->in_flight[slot] = false;
    ctx->slot = 0;
    return rc;
}

// ---- multi-GPU: communicators and group contexts -------------------------------------------------
static int comm_fail(drt_hip_ctx* ctx, const char* what, ncclResult_t r)
{
    if (ctx)
        ctx->err = std::string(what) + ": " + ncclGetErrorString(r);
    return DRT_ERR_COMM;
}

int drt_hip_comm_unique_id(drt_hip_unique_id* out)
{
    static_assert(sizeof(ncclUniqueId) == DRT_HIP_UNIQUE_ID_BYTES, "drt_hip_unique_id is an ncclUniqueId");
    if (!out)
        return DRT_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess)
        return DRT_ERR_COMM;
    memcpy(out->bytes, &id, sizeof id);
    return DRT_OK;
}

int drt_hip_comm_init_rank(drt_hip_ctx* ctx, const drt_hip_unique_id* id, int rank, int n_ranks)
{
    if (!ctx || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return ctx ? fail(ctx, DRT_ERR_INVALID, "comm_init_rank: bad arguments") : DRT_ERR_INVALID;
    if (!ctx->members.empty())
        return fail(ctx, DRT_ERR_INVALID, "comm_init_rank: a group context owns its communicators already");
    if (ctx->comm)
        return fail(ctx, DRT_ERR_INVALID, "comm_init_rank: the context already has a communicator");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId nid;
    memcpy(&nid, id->bytes, sizeof nid);
    const ncclResult_t r = ncclCommInitRank(&ctx->comm, n_ranks, nid, rank);
    if (r != ncclSuccess) {
        ctx->comm = nullptr;
        return comm_fail(ctx, "ncclCommInitRank", r);
    }
    ctx->comm_rank = rank;
    ctx->comm_size = n_ranks;
    return DRT_OK;
}

int drt_hip_device_pci_bus_id(const drt_hip_ctx* ctx, int member, char* out, int capacity)
{
    if (!ctx || !out || capacity < 16)
        return DRT_ERR_INVALID;
    const drt_hip_ctx* c = ctx;
    if (!ctx->members.empty()) {
        if (member < 0 || member >= (int)ctx->members.size())
            return DRT_ERR_INVALID;
        c = ctx->members[(size_t)member];
    } else if (member != 0)
        return DRT_ERR_INVALID;
    out[0] = 0;
    return hipDeviceGetPCIBusId(out, capacity, c->device) == hipSuccess ? DRT_OK : DRT_ERR_HIP;
}

int drt_hip_comm_size(const drt_hip_ctx* ctx)
{
    if (!ctx)
        return 0;
    if (!ctx->members.empty())
        return ctx->members[0]->comm_size;
    return ctx->comm ? ctx->comm_size : 0;
}

int drt_hip_comm_destroy(drt_hip_ctx* ctx)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream)
            (void)hipStreamSynchronize(ctx->stream);
        (void)ncclCommDestroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_size = 0;
    }
    return DRT_OK;
}

int drt_hip_create_group(const int* device_ids, int n_devices, drt_hip_ctx** out)
{
    if (!out)
        return DRT_ERR_INVALID;
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64)
        return DRT_ERR_INVALID;
    drt_hip_ctx* g = new drt_hip_ctx();
    g->device = device_ids[0];
    std::vector<int> distinct;           // the devices of the communicator, in order of first appearance
    for (int i = 0; i < n_devices; ++i) {
        drt_hip_ctx* m = nullptr;
        const int rc = drt_hip_create(device_ids[i], &m);
        if (rc != DRT_OK) {
            drt_hip_destroy(g);
            return rc;
        }
        m->is_member = true;
        g->members.push_back(m);
        int lead = i;
        for (int e = 0; e < i; ++e)
            if (device_ids[e] == device_ids[i]) { lead = e; break; }
        g->leader.push_back(lead);
        if (lead == i)
            distinct.push_back(device_ids[i]);
        if (hipEventCreateWithFlags(&m->ev_done, hipEventDisableTiming) != hipSuccess) {
            drt_hip_destroy(g);
            return DRT_ERR_HIP;
        }
    }
    std::vector<ncclComm_t> comms(distinct.size(), nullptr);
    const ncclResult_t cr = ncclCommInitAll(comms.data(), (int)distinct.size(), distinct.data());
    if (cr != ncclSuccess) {
        // (no context to carry the message: the caller gets the status, the RCCL text goes to stderr)
        fprintf(stderr, "[drt_hip] drt_hip_create_group: ncclCommInitAll over %zu devices failed: %s\n", distinct.size(), ncclGetErrorString(cr));
        for (ncclComm_t c : comms)
            if (c)
                (void)ncclCommAbort(c);
        drt_hip_destroy(g);             // (destroys the members created so far; none of them holds a communicator yet)
        return DRT_ERR_COMM;
    }
    for (int i = 0, k = 0; i < n_devices; ++i)
        if (g->leader[i] == i) {
            g->members[i]->comm = comms[k];
            g->members[i]->comm_rank = k;
            g->members[i]->comm_size = (int)distinct.size();
            ++k;
        }
    *out = g;
    return DRT_OK;
}

int drt_hip_group_size(const drt_hip_ctx* ctx) { return ctx ? (ctx->members.empty() ? 1 : (int)ctx->members.size()) : 0; }

int drt_hip_upload_scene(drt_hip_ctx* ctx, const drt_scene_desc* s)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (ctx->members.empty())
        return upload_scene_one(ctx, s);
    ctx->has_scene = false;
    for (drt_hip_ctx* m : ctx->members) {
        const int rc = upload_scene_one(m, s);
        if (rc != DRT_OK) {
            ctx->err = m->err;
            return rc;
        }
    }
    ctx->has_scene = true;
    return DRT_OK;
}

int drt_hip_set_specialisation(drt_hip_ctx* ctx, int mode)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (mode < DRT_SPECIALISE_GENERIC || mode > DRT_SPECIALISE_NOW)
        return fail(ctx, DRT_ERR_INVALID, "set_specialisation: unknown mode");
    ctx->jit_mode = mode;
    for (drt_hip_ctx* m : ctx->members)
        m->jit_mode = mode;
    return DRT_OK;
}

int drt_hip_update_params(drt_hip_ctx* ctx, const double* params)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (ctx->members.empty())
        return update_params_one(ctx, params);
    for (drt_hip_ctx* m : ctx->members) {
        const int rc = update_params_one(m, params);
        if (rc != DRT_OK) {
            ctx->err = m->err;
            return rc;
        }
    }
    return DRT_OK;
}

void* drt_hip_stream(drt_hip_ctx* ctx)
{
    if (!ctx)
        return nullptr;
    return (void*)(ctx->members.empty() ? ctx->stream : ctx->members[0]->stream);
}

int drt_hip_synchronize(drt_hip_ctx* ctx)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!ctx->members.empty()) {
        for (drt_hip_ctx* m : ctx->members) {
            const int rc = drt_hip_synchronize(m);
            if (rc != DRT_OK) {
                ctx->err = m->err;
                return rc;
            }
        }
        return DRT_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream)
        HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    return DRT_OK;
}

#ifdef DRT_BVH_STATS
// debug build only: read and clear the traversal counters of k_intersect_mesh (drt_kernels.h)
extern "C" int drt_hip_debug_bvh_stats(unsigned long long* out16)
{
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bvh_stats), 16 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    unsigned long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_stats), zero, sizeof zero);
    return 0;
}
#endif

#ifdef DRT_BVH_STATS
extern "C" int drt_hip_debug_bvh_hist(unsigned long long* out24)
{
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_bvh_hist), 24 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    unsigned long long zero[24] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_hist), zero, sizeof zero);
    return 0;
}
#endif
#ifdef DRT_WALK_TIMES
// debug build only: start / counters-dry / exit time of every wave of the last k_intersect_mesh launch (drt_kernels.h)
extern "C" int drt_hip_debug_walk_times(unsigned long long* out, int n_waves)
{
    (void)hipDeviceSynchronize();
    if (n_waves > 8192) n_waves = 8192;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_times), (size_t)n_waves * 3 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// hiprtc needs no device: the build container checks that the embedded headers still compile under it (tests/test_abi.py).
// Returns the size of the code object, or a negative status with the compiler's output in `log`.
extern "C" int drt_hip_debug_jit_compile(const char* arch, const char* name_expr, double* ms, char* log, int log_cap)
{
    if (!arch || !name_expr)
        return DRT_ERR_INVALID;
    const drt_jit::EntryPtr e = drt_jit::compile(arch, name_expr);
    const drt_jit::Code& c = e->code;
    if (ms)
        *ms = c.ms;
    if (log && log_cap > 0) {
        strncpy(log, c.log.c_str(), (size_t)log_cap - 1);
        log[log_cap - 1] = 0;
    }
    return c.ok ? (int)c.bin.size() : DRT_ERR_UNSUPPORTED;
}

const char* drt_hip_last_error(drt_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char* drt_hip_kernel_name(int k)
{
    static const char* names[DRT_K_COUNT] = {"k_raygen", "k_intersect", "k_shade", "k_film",
                                             "k_backward", "k_gradreduce", "k_intersect_mesh", "k_path"};
    return (k >= 0 && k < DRT_K_COUNT) ? names[k] : "";
}

} // extern "C"

