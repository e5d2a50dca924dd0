// drt_ctx.h -- host side of libdrt_hip.so: the context (one gfx950 device, its streams, its device buffers), the state of
// one render call between its phases, and the small helpers every part of the host runtime uses.  Included by
// drt_hip.hip only (one translation unit: the kernels are templates in headers and are instantiated where they are launched).
#pragma once

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct TimedLaunch {
    int kernel;
    hipEvent_t e0, e1;
};

// One render call between its phases: launch (everything enqueued, gradients in ctx->grad[ctx->slot]) -> reduce (the
// cross-device sum) -> collect (results on their way to the caller) -> finish (wait, hand over, statistics).
// A plain context runs them back to back; a group context runs each phase on ALL members before the next,
// so the devices work concurrently under one host thread.
struct RenderJob {
    drt_camera_desc cam;
    drt_render_params rp;
    const float* adjoint_rgb = nullptr;
    float* out_rgb = nullptr;
    double* out_param_grad = nullptr;
    float* out_gimg = nullptr;
    drt_hip_stats* stats = nullptr;
    int gimg_param = -1;
    bool backward = false, dev_out = false, timing = false, want_segments = false, sync = true;
    bool zero_copy = false;               // the image is written straight into the pinned block (synchronous host-buffer renders; drt_hip_render_async, one-stream form)
    bool direct_out = false, direct_gimg = false;   // ... or into the caller's own buffer, pinned with drt_hip_pin_host: nothing to hand over
    uint64_t done_seq = 0;                // the value the frame's last launch stores into the pinned block's completion word (0: none)
    bool copy_kernel = false;             // image, gradients and totals go to the pinned block by ONE launch on the copy stream
    int n_shards = 1, shard = 0, band = 1;
    uint32_t n_local_pixels = 0;
    size_t n_count_words = 0;
    float* d_out = nullptr;
    float* d_gimg = nullptr;
    size_t off_grad = 0, off_img = 0, off_gimg = 0, off_adj = 0, img_bytes = 0, grad_bytes = 0;
    drt_hip_stats st;
    std::chrono::steady_clock::time_point t0;
};

} // namespace

struct drt_hip_ctx {
    int device = 0;
    int n_cu = 256;
    uint64_t device_mem = (uint64_t)288 << 30;   // bytes of HBM (hipDeviceProp_t::totalGlobalMem)
    int mesh_blocks_per_cu = 4;           // resident blocks of k_intersect_mesh per CU (occupancy query): its persistent grid
    hipStream_t stream = nullptr;
    std::string err;

    bool has_scene = false;
    bool has_specular = false;
    bool emissive_bxdf = false;           // some analytic shape carries a BxDF AND an emitter (several emission terms per path)
    int max_colour_param = -1;            // largest parameter index that is some material's colour (device numbering)
    bool prog_ok = false;                 // k_path's intersection program covers the scene (drt_path.h)
    bool prog_sorted = false;             // the kind-sorted program covers the scene's analytic shapes (k_shade's tail)
    unsigned long long prog_sig[4] = {0, 0, 0, 0};   // the kinds of the scene's shapes, 3 bits each, 16 per word (KindSig, drt_prog.h)
    // run-time specialisation of k_path for this scene's shape kinds (drt_jit.h)
    std::string arch = "gfx950";          // hipDeviceProp_t::gcnArchName: what hiprtc compiles for
    int jit_mode = 1;                     // DRT_HIP_JIT: 0 = never, 1 = once the scene has rendered enough to pay for the compile, 2 ("force") = at once
    uint64_t scene_work = 0;              // path-bounces this scene has rendered through k_path (reset by upload_scene)
    std::map<std::string, hipFunction_t> jit_fn;   // instantiations loaded on this device, by name expression (nullptr: failed)
    std::vector<hipModule_t> jit_modules;
    std::string user_header;              // the scene's caller-defined shape kinds as the header hiprtc compiles them from (drt_prog.h); empty: none
    std::string jit_error;                // why the last specialisation failed (the kind-sorted program renders instead)
    double jit_ms = 0;                    // compile + load time spent by this context
    int n_params = 0, n_shapes = 0;   // n_params: as the device sees them (user parameters + internal constants)
    int n_user_params = 0;            // what the caller uploaded and gets gradients for
    int n_grad_slots = 0;             // how many of the first DRT_PATH_LDS_PARAMS parameters require a gradient (DevScene::grad_slot)
    std::vector<uint8_t> requires_grad;
    std::vector<drt_material_desc> materials;
    DevScene<float>* d_scene_f = nullptr;
    DevScene<double>* d_scene_d = nullptr;
    float* d_params_f = nullptr;
    double* d_params_d = nullptr;
    // triangle meshes (extension): one BVH over all triangles, in both compute types
    bool has_mesh = false;
    DevBvh<float> bvh_f{};
    DevBvh<double> bvh_d{};
    std::vector<void*> mesh_allocs;

    // FRAMES THAT OVERLAP.  A k_path grid ends with a last, partly filled round of waves, and the next frame's grid, on the same
    // stream, cannot start before it is over: 5-6 % of a fixed-depth frame, 25 % of a roulette-terminated one
    // (tools/two_frames.py).  Renders that do not wait for their results (device pointers without DRT_RENDER_SYNC) therefore
    // put the k_path launches of consecutive frames on TWO streams of their own, alternating, each with its own set of
    // partial-sum buffers; the finishing launch of every frame stays on the context's stream, in frame order, behind an
    // event -- what the caller sees (outputs written in stream order) does not change.
    hipStream_t path_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_begin[2] = {nullptr, nullptr}, ev_path[2] = {nullptr, nullptr};
    // k_path's partial-sum buffers come in two sets ("lanes": fpart/gpart/counts and fpart2/gpart2/counts2).  Whoever used a
    // lane last -- an overlapped frame or a plain render on the context's stream (lane 0) -- records ev_lane_free[lane] on the
    // context's stream once its last reader (the finishing launch) is enqueued; an overlapped k_path launch, which runs on a
    // stream of its own, waits for it before it writes the lane again.
    hipEvent_t ev_lane_free[2] = {nullptr, nullptr};
    bool lane_used[2] = {false, false};
    // drt_hip_update_params installs the new values with a launch on the context's stream and returns: ev_params marks the end of
    // that launch, params_pending[lane] says that lane's stream has not waited for it yet
    hipEvent_t ev_params = nullptr;
    bool params_pending[2] = {false, false};
    bool overlap_next = false;            // set around render_launch by the callers whose renders do not wait
    bool slot_used[DRT_HIP_FRAMES_IN_FLIGHT] = {};   // ev_copied[slot] has been recorded (the slot's buffers have a previous user)
    DevBuf fpart2, gpart2, counts2;       // k_path's partial sums of the odd frames
    DevBuf hist_ovf[2];                   // the one-launch kernels' general form: vertex-history words beyond the ones in LDS, one area per k_path stream
    DevBuf mesh_ovf[2];                   // k_path_mesh: traversal-stack entries beyond the ones in LDS, one area per k_path stream
    // (scenes with a mesh: three sets of queue lanes and hit lanes -- a shade launch reads depth k and appends to k + 1 and k + 2 --
    //  and two sets of candidate lists, by the parity of their depth)
    DevBuf ray_a[3], ray_b[3], ray_id[3], hit, hit2, hit3, lacc, gpath, gfilm, gimg_out, tape, nv, fpart, gpix, cand[2], cand_a[2], cand_b[2], cand_count[2],
        ch_cva, ch_cvb, ch_cvh, ch_nxa, ch_nxb, ch_nxh, ch_g, ch_w, ch_ids, ch_ndraw, ch_dbase, counts, segtotal[DRT_HIP_FRAMES_IN_FLIGHT], film, gpart, grad[DRT_HIP_FRAMES_IN_FLIGHT], adjoint, out[DRT_HIP_FRAMES_IN_FLIGHT];   // one set per frame in flight (drt_hip_render_async; device frames that do not wait alternate between the first two), slot 0 otherwise
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<TimedLaunch> timed;
    unsigned long long h_segments = 0;
    unsigned long long* h_probe = nullptr;   // pinned: queue-length polls of deep-cap renders
    double* h_params = nullptr;              // pinned: the values of the last drt_hip_update_params, read by the launch that installs them
    size_t h_params_cap = 0;
    // pinned staging of everything a host-buffer render returns: [segments 8 B | grads | image | gradient
    // image] arrive by DMA in one go, then plain memcpys into the caller's (pageable) buffers -- a
    // pageable hipMemcpy of the 3 MB image alone cost 1 ms
    uint8_t* h_stage[DRT_HIP_FRAMES_IN_FLIGHT] = {};
    size_t h_stage_cap[DRT_HIP_FRAMES_IN_FLIGHT] = {};
    // asynchronous host-buffer renders (drt_hip_render_async / drt_hip_wait): up to two frames in flight; frame t uses set
    // t & 1, its results travel to the pinned block on copy_stream while the next frame's kernels run on `stream`
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_rendered[DRT_HIP_FRAMES_IN_FLIGHT] = {}, ev_copied[DRT_HIP_FRAMES_IN_FLIGHT] = {};
    RenderJob pending[DRT_HIP_FRAMES_IN_FLIGHT];
    bool in_flight[DRT_HIP_FRAMES_IN_FLIGHT] = {};
    uint64_t next_ticket = 1;
    const void* adj_src_host = nullptr;   // a host-buffer render's adjoint image in pinned memory (render_launch), ...
    const void* adj_src_dev = nullptr;    // ... as the device addresses it; render_impl either hands this to the kernels or copies it
    size_t adj_bytes = 0;
    bool adj_pending = false;
    bool stage_adjoint_next = false;      // set around the render_launch of an asynchronous host-buffer frame: its adjoint image is copied before the call returns, pinned by the caller or not
    bool zero_copy_next = false;          // set around the render_launch of a host-buffer render whose finishing kernels store the image into the pinned block
    struct PinnedRange { uint8_t* host; size_t bytes; uint8_t* dev; };
    std::vector<PinnedRange> pinned;      // drt_hip_pin_host: caller buffers the finishing kernels may write directly
    uint64_t done_seq = 0;                // completion words handed out so far (render_collect)
    uint64_t dev_frames = 0;              // renders made with DRT_RENDER_ALLREDUCE_ASYNC (their gradient set alternates)
    int slot = 0;                         // which of the double-buffered sets (grad, segtotal, out, h_stage) this render uses
    DevBuf probe;
    uint64_t bvh_bytes = 0;
    RenderJob job;
    // multi-GPU.  A plain context may join a communicator (one process per GPU, drt_hip_comm_init_rank).  A GROUP
    // context (drt_hip_create_group) owns one plain member per listed device and nothing else; members that share a
    // device are summed on it, the first member of every distinct device (its "leader") holds that device's rank
    // in the group's communicator.
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    std::vector<drt_hip_ctx*> members;
    std::vector<int> leader;          // member i -> index of the first member on the same device
    hipEvent_t ev_done = nullptr;     // member: "my gradient is complete" (waited for by its leader's stream)
    bool is_member = false;
};

namespace {

#define HIPCHK(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return e_ == hipErrorOutOfMemory ? DRT_ERR_OOM : DRT_ERR_HIP;                     \
        }                                                                                     \
    } while (0)

int fail(drt_hip_ctx* ctx, int code, const char* msg)
{
    ctx->err = msg;
    return code;
}

int ensure(drt_hip_ctx* ctx, DevBuf& b, size_t bytes)
{
    if (bytes <= b.cap)
        return DRT_OK;
    if (b.p) {
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        e = hipMalloc(&b.p, want);
    }
    if (e != hipSuccess) {
        b.p = nullptr;
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return DRT_ERR_OOM;
    }
    b.cap = want;
    return DRT_OK;
}

void release(DevBuf& b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

void release_mesh(drt_hip_ctx* ctx)
{
    for (void* p : ctx->mesh_allocs)
        (void)hipFree(p);
    ctx->mesh_allocs.clear();
    ctx->has_mesh = false;
    memset(&ctx->bvh_f, 0, sizeof ctx->bvh_f);
    memset(&ctx->bvh_d, 0, sizeof ctx->bvh_d);
}


} // namespace
