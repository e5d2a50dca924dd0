// drt_hip.hip -- host runtime of libdrt_hip.so: the C ABI of include/drt_hip.h over the
// wavefront kernels of drt_kernels.h.  One context = one gfx950 device + one stream; every
// bounce is two launches (K2, K3) on persistent grids that read their queue length from device
// memory, so a whole render is enqueued without a single host round trip.
#include "drt_kernels.h"
#include "drt_path.h"
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
    bool zero_copy = false;               // the image is written straight into the pinned block (drt_hip_render_async, one-stream form)
    bool copy_kernel = false;             // image, gradients and totals go to the pinned block by ONE launch on the copy stream
    int n_shards = 1, shard = 0, band = 1;
    uint32_t n_local_pixels = 0;
    size_t n_count_words = 0;
    float* d_out = nullptr;
    float* d_gimg = nullptr;
    size_t off_grad = 0, off_img = 0, off_gimg = 0, img_bytes = 0, grad_bytes = 0;
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
    std::string jit_error;                // why the last specialisation failed (the kind-sorted program renders instead)
    double jit_ms = 0;                    // compile + load time spent by this context
    int n_params = 0, n_shapes = 0;   // n_params: as the device sees them (user parameters + internal constants)
    int n_user_params = 0;            // what the caller uploaded and gets gradients for
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
    bool overlap_next = false;            // set around render_launch by the callers whose renders do not wait
    bool slot_used[DRT_HIP_FRAMES_IN_FLIGHT] = {};   // ev_copied[slot] has been recorded (the slot's buffers have a previous user)
    DevBuf fpart2, gpart2, counts2;       // k_path's partial sums of the odd frames
    DevBuf ray_a[2], ray_b[2], ray_id[2], hit, hit2, lacc, gpath, gfilm, gimg_out, tape, nv, fpart, gpix, cand, cand_a, cand_b, cand_count,
        ch_cva, ch_cvb, ch_cvh, ch_nxa, ch_nxb, ch_nxh, ch_g, ch_w, ch_ids, ch_ndraw, ch_dbase, counts, segtotal[DRT_HIP_FRAMES_IN_FLIGHT], film, gpart, grad[DRT_HIP_FRAMES_IN_FLIGHT], adjoint, out[DRT_HIP_FRAMES_IN_FLIGHT];   // one set per frame in flight (drt_hip_render_async; device frames that do not wait alternate between the first two), slot 0 otherwise
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;
    std::vector<TimedLaunch> timed;
    unsigned long long h_segments = 0;
    unsigned long long* h_probe = nullptr;   // pinned: queue-length polls of deep-cap renders
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
    bool zero_copy_next = false;          // set around the render_launch of an asynchronous host-buffer render
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

template <typename R4>
int upload_array(drt_hip_ctx* ctx, const std::vector<R4>& host, const R4** dev)
{
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, host.empty() ? 16 : host.size() * sizeof(R4));
    if (e == hipSuccess && !host.empty())
        e = hipMemcpy(p, host.data(), host.size() * sizeof(R4), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        ctx->err = std::string("mesh upload: ") + hipGetErrorString(e);
        if (p) (void)hipFree(p);
        return DRT_ERR_HIP;
    }
    ctx->mesh_allocs.push_back(p);
    *dev = (const R4*)p;
    return DRT_OK;
}

inline float link_bits(float, uint32_t v) { float f; memcpy(&f, &v, 4); return f; }
inline double link_bits(double, uint32_t v) { return (double)v; }

// device image of the BVH in compute type R
template <typename R>
int upload_bvh(drt_hip_ctx* ctx, const drt_bvh::Built& b, const std::vector<drt_bvh::Tri>& tris, DevBvh<R>* out)
{
    typedef typename Q4<R>::T R4;
    std::vector<uint4> nodes(b.nodes.size() * 4);
    for (size_t i = 0; i < b.nodes.size(); ++i) {
        const drt_bvh::QNode q = drt_bvh::quantise(b.nodes[i]);
        memcpy(&nodes[i * 4], q.w, sizeof q.w);
    }
    std::vector<R4> ta(b.order.size()), tb(b.order.size()), tc(b.order.size()), ts(tris.size());
    for (size_t j = 0; j < b.order.size(); ++j) {
        const drt_bvh::Tri& t = tris[b.order[j]];
        ta[j].x = (R)t.v0[0]; ta[j].y = (R)t.v0[1]; ta[j].z = (R)t.v0[2]; ta[j].w = (R)t.e1[0];
        tb[j].x = (R)t.e1[1]; tb[j].y = (R)t.e1[2]; tb[j].z = (R)t.e2[0]; tb[j].w = (R)t.e2[1];
        tc[j].x = (R)t.e2[2]; tc[j].y = link_bits(R(0), t.global); tc[j].z = link_bits(R(0), t.flat); tc[j].w = R(0);
    }
    for (size_t g = 0; g < tris.size(); ++g) {
        const drt_bvh::Tri& t = tris[g];
        ts[t.global].x = (R)t.n[0]; ts[t.global].y = (R)t.n[1]; ts[t.global].z = (R)t.n[2];
        ts[t.global].w = link_bits(R(0), t.ids);
    }
    int rc;
    if ((rc = upload_array<uint4>(ctx, nodes, &out->node)) != DRT_OK) return rc;
    {   // one record of three 16-byte words per triangle: a leaf's triangles are one or two cache lines, not three
        std::vector<R4> t3(ta.size() * 3);
        for (size_t j = 0; j < ta.size(); ++j) { t3[j * 3] = ta[j]; t3[j * 3 + 1] = tb[j]; t3[j * 3 + 2] = tc[j]; }
        if ((rc = upload_array(ctx, t3, &out->tri)) != DRT_OK) return rc;
    }
    if ((rc = upload_array(ctx, ts, &out->tri_shade)) != DRT_OK) return rc;
    out->n_nodes = (uint32_t)b.nodes.size();
    out->n_top = b.top;
    out->n_tris = (uint32_t)tris.size();
    // the box around everything, as the root's (padded) child boxes give it, rounded outwards in R
    for (int a = 0; a < 3; ++a) {
        double lo = INFINITY, hi = -INFINITY;
        for (int c = 0; c < drt_bvh::kWidth; ++c)
            if (!b.nodes.empty() && b.nodes[0].child[c] != drt_bvh::kLeaf) {
                lo = std::min(lo, b.nodes[0].lo[c][a]);
                hi = std::max(hi, b.nodes[0].hi[c][a]);
            }
        R rl = (R)lo, rh = (R)hi;
        if ((double)rl > lo) rl = std::nextafter(rl, (R)-INFINITY);
        if ((double)rh < hi) rh = std::nextafter(rh, (R)INFINITY);
        out->lo[a] = rl;
        out->hi[a] = rh;
    }
    return DRT_OK;
}

template <typename R>
void fill_scene(DevScene<R>& ds, std::vector<R>& params, const drt_scene_desc* s, unsigned long long sig[4])
{
    sig[0] = sig[1] = sig[2] = sig[3] = 0;
    memset(&ds, 0, sizeof ds);
    ds.n_shapes = s->n_shapes;
    ds.n_materials = s->n_materials;
    ds.n_emitters = s->n_emitters;
    // a mirror has no colour parameter (bxdf.hpp:126-144): its materials point at an internal constant
    // (1, 1, 1) appended after the caller's parameters (never reported, never differentiated)
    bool any_mirror = false;
    for (int i = 0; i < s->n_materials; ++i)
        any_mirror = any_mirror || s->materials[i].type == DRT_BXDF_MIRROR;
    ds.n_params = s->n_params + (any_mirror ? 1 : 0);
    int flat = 0;
    for (int i = 0; i < s->n_shapes; ++i) {
        ds.flat[i] = flat;
        flat += s->shapes[i].type == DRT_SHAPE_MESH ? s->meshes[s->shapes[i].mesh].n_triangles : 1;
        for (int j = 0; j < 4; ++j)
            ds.shapes[i].p[j] = (R)s->shapes[i].p[j];
        ds.shapes[i].type = s->shapes[i].type;
        if (s->shapes[i].type == DRT_SHAPE_PLANE) ds.plane_mask |= 1ull << i;
        if (s->shapes[i].type == DRT_SHAPE_SPHERE) ds.sphere_mask |= 1ull << i;
        ds.shapes[i].material = s->shapes[i].material;
        ds.shapes[i].emitter = s->shapes[i].emitter;
    }
    // the intersection program of the packed f32 test: scene order, adjacent planes / spheres paired
    for (int i = 0; i < s->n_shapes;) {
        const int t = s->shapes[i].type;
        const int it = ds.n_items++;
        if (t == DRT_SHAPE_MESH) {
            ds.item_skip |= 1ull << it;
            i += 1;
            continue;
        }
        if (t == DRT_SHAPE_SPHERE)
            ds.item_sphere |= 1ull << it;
        if (i + 1 < s->n_shapes && s->shapes[i + 1].type == t) {
            ds.item_pair |= 1ull << it;
            for (int j = 0; j < 4; ++j) {
                ds.items[it][2 * j] = (R)s->shapes[i].p[j];
                ds.items[it][2 * j + 1] = (R)s->shapes[i + 1].p[j];
            }
            i += 2;
        } else {
            for (int j = 0; j < 4; ++j)
                ds.items[it][j] = (R)s->shapes[i].p[j];
            i += 1;
        }
    }
    for (int i = 0; i < s->n_materials; ++i) {
        ds.materials[i].type = s->materials[i].type;
        ds.materials[i].param = s->materials[i].type == DRT_BXDF_MIRROR ? s->n_params : s->materials[i].param;
        ds.materials[i].exponent = (R)s->materials[i].exponent;
        ds.materials[i].norm = (R)((s->materials[i].exponent + 2.0) / (2.0 * DRT_PI));
    }
    for (int i = 0; i < s->n_emitters; ++i)
        ds.emitter_param[i] = s->emitters[i].param;
    // k_path (drt_path.h): the parameter ids of every shape in one word, and the intersection program
    ds.prog_ok = 1;
    bool has_mesh_shape = false;
    int kinds[DRT_MAX_SHAPES];
    R recs[DRT_MAX_SHAPES][4];
    for (int i = 0; i < s->n_shapes; ++i) {
        const int m = s->shapes[i].material, e = s->shapes[i].emitter;
        const uint32_t cid = m >= 0 ? (uint32_t)ds.materials[m].param : DRT_ID_NONE;
        const uint32_t eid = e >= 0 ? (uint32_t)s->emitters[e].param : DRT_ID_NONE;
        ds.shapes[i].pad = (int)(cid | (eid << 16));
        kinds[i] = 7;                           // (a mesh record: belongs to no kind loop -- k_path does not walk meshes,
        recs[i][0] = recs[i][1] = recs[i][2] = recs[i][3] = R(0);   //  k_shade's tail tests the analytic rest)
        if (s->shapes[i].type == DRT_SHAPE_MESH) {
            has_mesh_shape = true;
        } else {
            int kind = s->shapes[i].type == DRT_SHAPE_SPHERE ? DRT_PK_SPHERE : DRT_PK_PLANE;
            R rec[4] = {(R)s->shapes[i].p[0], (R)s->shapes[i].p[1], (R)s->shapes[i].p[2], (R)s->shapes[i].p[3]};
            if (kind == DRT_PK_PLANE) {
                // n = +-e_a exactly: t = (sgn off - o_a) * rcp(d_a), bit-identical to the general form (drt_path.h)
                int axis = -1, nonzero = 0;
                for (int a = 0; a < 3; ++a)
                    if (rec[a] != R(0)) { ++nonzero; axis = a; }
                if (nonzero == 1 && (rec[axis] == R(1) || rec[axis] == R(-1))) {
                    kind = DRT_PK_AX + axis;
                    rec[0] = rec[axis] * rec[3];
                    rec[1] = rec[2] = rec[3] = R(0);
                }
            }
            kinds[i] = kind;
            for (int j = 0; j < 4; ++j)
                recs[i][j] = rec[j];
        }
        sig[i >> 4] |= (unsigned long long)kinds[i] << (3 * (i & 15));   // the scene's signature (KindSig, drt_prog.h)
    }
    {   // the kind-sorted copy (stable: scene order inside a kind)
        int n = 0;
        for (int k = 0; k < 5; ++k) {
            ds.kind_begin[k] = n;
            for (int i = 0; i < s->n_shapes; ++i)
                if (kinds[i] == k) {
                    for (int j = 0; j < 4; ++j)
                        ds.sorted[n][j] = recs[i][j];
                    ds.sorted_shape[n++] = i;
                }
        }
        for (int k = 5; k < 8; ++k)
            ds.kind_begin[k] = n;
    }
    ds.prog_sorted = ds.prog_ok;                   // the sorted program is valid (for the analytic shapes)
    if (has_mesh_shape)
        ds.prog_ok = 0;                         // ... but k_path is not for scenes with a mesh
    params.assign((size_t)ds.n_params * 3, R(1));
    for (size_t i = 0; i < (size_t)s->n_params * 3; ++i)
        params[i] = (R)s->params[i];
}

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

// K2's grid in a scene with a mesh: every wave of k_intersect leaves ONE candidate list, and the BVH walk's waves pull
// whole lists -- so the lists should be short (a wave that pulls the last list works on it alone: measured on
// 512 x 512 x 64, 550-ray lists: 6.2 ms per step in the walk, 137-ray lists: 5.1 ms) but not nearly empty either
// (every pull is an atomic round trip): ~8 chunks of 64 rays per wave, a quarter of them candidates.
int k2_mesh_grid(const drt_hip_ctx* ctx, uint64_t n_paths)
{
    static const int per_block = getenv("DRT_HIP_K2_PATHS_PER_BLOCK") ? atoi(getenv("DRT_HIP_K2_PATHS_PER_BLOCK")) : 2048;
    uint64_t blocks = (n_paths + (uint64_t)per_block - 1) / (uint64_t)per_block;
    static const int max_per_cu = getenv("DRT_HIP_K2_MAX_BLOCKS_PER_CU") ? atoi(getenv("DRT_HIP_K2_MAX_BLOCKS_PER_CU")) : 32;
    const uint64_t lo = (uint64_t)ctx->n_cu, hi = (uint64_t)ctx->n_cu * (uint64_t)max_per_cu;
    if (blocks < lo) blocks = lo;
    if (blocks > hi) blocks = hi;
    return (int)blocks;
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
// the kind-sorted program, same results).
hipFunction_t jit_function(drt_hip_ctx* ctx, const std::string& name_expr)
{
    auto it = ctx->jit_fn.find(name_expr);
    if (it != ctx->jit_fn.end())
        return it->second;
    const auto t0 = std::chrono::steady_clock::now();
    hipFunction_t fn = nullptr;
    const drt_jit::Code& c = drt_jit::compile(ctx->arch, name_expr);
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
    if (!fn && getenv("DRT_HIP_JIT_VERBOSE"))
        fprintf(stderr, "[drt_hip] specialisation failed: %s\n", ctx->jit_error.c_str());
    ctx->jit_fn[name_expr] = fn;
    ctx->jit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return fn;
}
// A compile costs ~0.5 s of host time and buys ~25 % of the kind-sorted program's time: it pays once the scene has rendered
// a few seconds' worth of frames.  2^31 path-bounces are ~20 ms of rendering: small test frames never get there, a bench or an
// optimisation loop does within its first frames.
#define DRT_JIT_AFTER_WORK ((uint64_t)1 << 31)

#define DRT_POLL_EVERY 4
#define DRT_TOTAL_WORDS 8           // segtotal: segments, queue rays read, written, capped paths, K2 rays, walked candidates

template <typename R>
int render_impl(drt_hip_ctx* ctx, const drt_camera_desc* cam, const drt_render_params* rp,
                const float* d_adjoint, float* d_out_rgb, bool backward, bool timing,
                drt_hip_stats* st, uint32_t n_local_pixels, int depth_cap, size_t* n_count_words,
                double* film, int gimg_param = -1, double* gfilm = nullptr, float* d_out_gimg = nullptr)
{
    typedef typename Q4<R>::T R4;
    const DevScene<R>* d_scene = sizeof(R) == 4 ? (const DevScene<R>*)ctx->d_scene_f
                                                : (const DevScene<R>*)ctx->d_scene_d;
    const R* d_params = sizeof(R) == 4 ? (const R*)ctx->d_params_f : (const R*)ctx->d_params_d;
    DevBvh<R> bvh;
    memcpy(&bvh, sizeof(R) == 4 ? (const void*)&ctx->bvh_f : (const void*)&ctx->bvh_d, sizeof bvh);
    const int spp = rp->spp;
    const uint64_t total_paths = (uint64_t)n_local_pixels * (uint64_t)spp;
    const int D = depth_cap;
    const bool unbiased = backward && (rp->flags & DRT_RENDER_UNBIASED) != 0 && gimg_param < 0;
    // K2 folded into K3 wherever nothing else consumes the hit records: never with a mesh (the BVH
    // walk is its own kernel); in the unbiased backward every depth but the first of a chain, whose
    // hit is saved as the next chain vertex
    static const bool fuse_env = !(getenv("DRT_HIP_FUSE") && atoi(getenv("DRT_HIP_FUSE")) == 0);
    const bool can_fuse = fuse_env && !ctx->has_mesh;
    static const int shade_nb_env = getenv("DRT_HIP_SHADE_BOUNCES") ? atoi(getenv("DRT_HIP_SHADE_BOUNCES")) : 0;
    // ---- k_path (drt_path.h): the whole path in one launch, in registers.  Taken when the scene is analytic,
    // the estimator the biased one and at most 4 parameters want gradients.  Two forms: lanes in lockstep (all at
    // the same depth; a lane whose path ended idles to the end of the sample) when most lanes stay busy to the end --
    // ~7 % of the paths end per bounce on a miss or a light, the roulette removes `absorb` of the rest from
    // min_bounces on -- and the regenerating form (a lane whose path ended starts its next sample at once) otherwise:
    // roulette-terminated paths under the default cap of 64, the reference's own defaults (-b 1 -p 0.5).
    static const int path_env = getenv("DRT_HIP_PATH") ? atoi(getenv("DRT_HIP_PATH")) : 1;
    static const bool path_unb_env = !(getenv("DRT_HIP_PATH_UNBIASED") && atoi(getenv("DRT_HIP_PATH_UNBIASED")) == 0);
    bool use_path = path_env > 0 && can_fuse && ctx->prog_ok && (!unbiased || path_unb_env) && D > 0 &&
                    (!(backward || gimg_param >= 0) || ctx->n_params <= DRT_FAST_PARAMS) && rp->bounces_per_launch <= 0 && shade_nb_env <= 0 &&
                    !getenv("DRT_HIP_DUMP_PATH");
    static const int regen_env = getenv("DRT_HIP_PATH_REGEN") ? atoi(getenv("DRT_HIP_PATH_REGEN")) : -1;
    bool path_regen = regen_env > 0;
    if (unbiased)
        path_regen = false;                    // (k_path_unbiased walks its samples in lockstep)
    if (use_path && regen_env < 0 && !unbiased) {
        // lockstep: a wave runs until the longest of its 64 paths ends -- the depth cap for fixed-depth renders, under the
        // roulette about the depth that 1 path in 256 reaches; regenerating: every lane runs the mean path length, at
        // ~1.7 x the cost per bounce (per-lane depth bookkeeping, 126-139 registers) + the camera code inside the loop.
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
        path_regen = 1.7 * mean_len + 0.5 < (double)longest;
    }
    // Batch = the paths that are in flight at once on the queue route.  The BVH walk wants it LARGE: its launches end in a
    // tail of ~0.1 ms whatever their size (the list counters run dry, every wave finishes what it holds), so config 4 at full
    // size (1024^2 x 256 spp) takes 115 / 101 / 98 / 96 ms with 2^24 / 2^26 / 2^27 / 2^28 paths per batch and one GPU's
    // share of it (33.5 M paths) 14.3 ms in two batches, 13.1 in one.  Every path in flight owns ~0.2 KB of queue lanes,
    // tape and candidate records (twice that in f64): the default is the largest power of two whose buffers fit in an eighth
    // of the device's memory, at most 32 GB -- 2^27 paths (27 GB) for a depth-8 f32 render on a 288 GB part.
    uint64_t cap_default = (uint64_t)1 << 24;
    {
        const uint64_t f = sizeof(R) / 4;
        const uint64_t per_path = f * (112u + 8u * (uint64_t)(D > 0 ? D : 1) + (ctx->has_mesh ? 36u : 0u) + (unbiased ? 110u : 0u)) + 24u;
        const uint64_t budget = std::min<uint64_t>(ctx->device_mem / 8, (uint64_t)32 << 30);
        cap_default = (uint64_t)1 << 22;
        while (cap_default < ((uint64_t)1 << 28) && 2 * cap_default * per_path <= budget)
            cap_default *= 2;
        // (a device that other work has filled: no more than half of what is free now, unless the buffers exist already)
        if (!use_path && rp->batch_paths <= 0 && (uint64_t)ctx->ray_a[0].cap < std::min<uint64_t>(cap_default, total_paths) * 16u * f) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                while (cap_default > ((uint64_t)1 << 22) && cap_default * per_path > (uint64_t)free_b / 2)
                    cap_default /= 2;
        }
    }
    uint64_t cap = rp->batch_paths > 0 ? (uint64_t)rp->batch_paths : cap_default;
    if (use_path && rp->batch_paths <= 0)
        cap = total_paths;                 // no per-path memory: one batch covers the frame
    if (const char* e = getenv("DRT_HIP_BATCH_PATHS")) {
        long long v = atoll(e);
        if (v > 0)
            cap = (uint64_t)v;
    }
    if (cap > total_paths)
        cap = total_paths;
    if (cap < 1)
        cap = 1;
    if (cap > 0x7FFFFFFFull)
        cap = 0x7FFFFFFFull;
    uint32_t Pb = (uint32_t)(cap / (uint64_t)spp);
    if (Pb < 1) Pb = 1;
    if (Pb > n_local_pixels) Pb = n_local_pixels;
    uint32_t Sb = (uint32_t)(cap / Pb);
    if (Sb > (uint32_t)spp) Sb = (uint32_t)spp;
    if (Sb < 1) Sb = 1;
    const size_t N = (size_t)Pb * Sb;   // batch capacity in paths
    // queue regions: one wave each; enough of them to fill 256 CUs several times over
    uint32_t region_shift = 8;   // 256 slots: 4 chunks per wave (sweep in profiles/: 64..4096)
    if (const char* e = getenv("DRT_HIP_REGION_SIZE")) {
        long v = atol(e);
        for (region_shift = 6; region_shift < 20 && (1l << region_shift) < v; ++region_shift) { }
    }
    while (region_shift > 6 && (N >> region_shift) < (size_t)ctx->n_cu * 32)
        --region_shift;
    const uint32_t region_size = 1u << region_shift;
    const uint32_t max_regions = (uint32_t)((N + region_size - 1) / region_size);
    // k_path geometry: wave <-> (64 pixels, spr samples); enough waves for ~5-6 rounds of the 5,120 the chip holds (the
    // tail stays short) in ranges of equal length (sweep on config 3, ms per launch: 16 samples per range 0.827,
    // 13: 0.818, 10: 0.793, 8: 0.812, 7: 0.795, 4: 0.811)
    const uint32_t path_groups = (Pb + DRT_WAVE - 1) / DRT_WAVE;
    uint32_t path_spr = 1;
    {
        // (regenerating lanes balance themselves over their sample range: longer ranges, fewer waves)
        const uint64_t target = (uint64_t)ctx->n_cu * (path_regen ? 32 : 112);
        const uint64_t want = std::max<uint64_t>(1, (target + path_groups - 1) / path_groups);   // ranges
        path_spr = (uint32_t)((Sb + want - 1) / want);
        if (const char* e = getenv("DRT_HIP_PATH_SPR"))
            path_spr = (uint32_t)atoi(e);
        if (path_spr < 1) path_spr = 1;
        if (path_spr > Sb) path_spr = Sb;
    }
    const uint32_t path_ranges = (Sb + path_spr - 1) / path_spr;
    const size_t path_waves = (size_t)path_groups * path_ranges;

    // the rays the BVH walk has to see: one dense list per wave of k_intersect's persistent grid (+ the walk's list counter)
    // (list l = span [l * cand_cap, ...) of `cand`, cand_cap = the chunks one K2 wave of THIS launch sees, x 64)
    const uint32_t k2_waves = (uint32_t)k2_mesh_grid(ctx, N) * (DRT_BLOCK / DRT_WAVE);
    const size_t cand_words = (((size_t)max_regions << (region_shift - 6)) + k2_waves) * DRT_WAVE;   // (>= max_regions * region_size)
    // scenes with a mesh: the shade launch intersects the ray it produces with the analytic shapes and builds the BVH
    // walk's candidate lists itself (k_shade<TAIL>); the hit lane is double-buffered like the queue
    static const uint32_t shade_list_group = getenv("DRT_HIP_SHADE_LIST_GROUP") ? (uint32_t)std::max(1, atoi(getenv("DRT_HIP_SHADE_LIST_GROUP"))) : 4u;
    static const bool tail_env = !(getenv("DRT_HIP_SHADE_TAIL") && atoi(getenv("DRT_HIP_SHADE_TAIL")) == 0);
    const bool shade_tail = tail_env && ctx->has_mesh && !can_fuse && (ctx->prog_sorted || sizeof(R) == 8);
    int rc;
    ChainState<R> cs;
    memset(&cs, 0, sizeof cs);
    size_t cw = (size_t)(D + 1) * max_regions;   // counts[depth][region] of one batch
    const bool overlap_ok = ctx->overlap_next && use_path && !timing && gimg_param < 0 && ctx->path_stream[0] && ctx->ev_copied[0];
    DevBuf& fpart_buf = overlap_ok && (ctx->slot & 1) ? ctx->fpart2 : ctx->fpart;
    DevBuf& gpart_buf = overlap_ok && (ctx->slot & 1) ? ctx->gpart2 : ctx->gpart;
    DevBuf& counts_buf = overlap_ok && (ctx->slot & 1) ? ctx->counts2 : ctx->counts;
    if (use_path) {
        cw = 2 * path_waves;                     // [segments | capped paths] per wave
        if ((rc = ensure(ctx, fpart_buf, (size_t)path_ranges * 3 * Pb * sizeof(double))) != DRT_OK) return rc;
        if (gimg_param >= 0)
            if ((rc = ensure(ctx, ctx->gpix, (size_t)path_ranges * 3 * Pb * sizeof(double))) != DRT_OK) return rc;
    } else {
    for (int i = 0; i < 2; ++i) {
        if ((rc = ensure(ctx, ctx->ray_a[i], N * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ray_b[i], N * sizeof(typename Q2<R>::T))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ray_id[i], N * sizeof(uint2))) != DRT_OK) return rc;
    }
    if ((rc = ensure(ctx, ctx->hit, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
    if (shade_tail)
        if ((rc = ensure(ctx, ctx->hit2, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
    if (ctx->has_mesh) {
        if ((rc = ensure(ctx, ctx->cand, cand_words * sizeof(uint32_t))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->cand_a, cand_words * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->cand_b, cand_words * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->cand_count, ((size_t)std::max<uint32_t>(k2_waves, max_regions) + DRT_PULL_WORDS) * sizeof(uint32_t))) != DRT_OK) return rc;
    }
    if ((rc = ensure(ctx, ctx->lacc, N * sizeof(R4))) != DRT_OK) return rc;
    if (gimg_param >= 0)
        if ((rc = ensure(ctx, ctx->gpath, N * sizeof(R4))) != DRT_OK) return rc;
    if (unbiased) {
        typedef typename Q2<R>::T R2c;
        if ((rc = ensure(ctx, ctx->ch_cva, N * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_cvb, N * sizeof(R2c))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_cvh, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_nxa, N * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_nxb, N * sizeof(R2c))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_nxh, N * sizeof(HitRec<R>))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_g, N * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_w, N * sizeof(R4))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_ids, N * sizeof(uint32_t))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_ndraw, N * sizeof(uint32_t))) != DRT_OK) return rc;
        if ((rc = ensure(ctx, ctx->ch_dbase, N * sizeof(uint32_t))) != DRT_OK) return rc;
        cs.cv_a = (R4*)ctx->ch_cva.p; cs.cv_b = (R2c*)ctx->ch_cvb.p; cs.cv_hit = (HitRec<R>*)ctx->ch_cvh.p;
        cs.nx_a = (R4*)ctx->ch_nxa.p; cs.nx_b = (R2c*)ctx->ch_nxb.p; cs.nx_hit = (HitRec<R>*)ctx->ch_nxh.p;
        cs.g = (R4*)ctx->ch_g.p; cs.w = (R4*)ctx->ch_w.p;
        cs.ids = (uint32_t*)ctx->ch_ids.p; cs.ndraw = (uint32_t*)ctx->ch_ndraw.p; cs.dbase = (uint32_t*)ctx->ch_dbase.p;
    }
    if ((rc = ensure(ctx, ctx->tape, N * sizeof(TapeRec<R>) * (size_t)(D > 0 ? D : 1))) != DRT_OK) return rc;
    if ((rc = ensure(ctx, ctx->nv, N * sizeof(uint32_t))) != DRT_OK) return rc;
    }
    *n_count_words = cw;
    if ((rc = ensure(ctx, counts_buf, cw * sizeof(uint32_t))) != DRT_OK) return rc;
    if ((rc = ensure(ctx, ctx->segtotal[ctx->slot], DRT_TOTAL_WORDS * sizeof(unsigned long long))) != DRT_OK) return rc;   // segments, queue rays read, written, capped, K2 rays, walked candidates
    // a k_path launch that covers the whole frame is followed by ONE finishing launch that WRITES image, gradients and
    // totals (k_path_finish); every other route accumulates into zeroed buffers
    static const bool finish_env = !(getenv("DRT_HIP_PATH_FINISH") && atoi(getenv("DRT_HIP_PATH_FINISH")) == 0);
    const bool path_finish = finish_env && use_path && Pb == n_local_pixels && Sb == (uint32_t)spp && (!film || d_out_rgb);
    if (!path_finish) {
        HIPCHK(ctx, hipMemsetAsync(ctx->segtotal[ctx->slot].p, 0, DRT_TOTAL_WORDS * sizeof(unsigned long long), ctx->stream));
        if (film)
            HIPCHK(ctx, hipMemsetAsync(film, 0, (size_t)n_local_pixels * 3 * sizeof(double), ctx->stream));
        if (backward)
            HIPCHK(ctx, hipMemsetAsync(ctx->grad[ctx->slot].p, 0, (size_t)(ctx->n_params ? ctx->n_params : 1) * 3 * sizeof(double), ctx->stream));
    }
    const int bwd_grid = grid_for(ctx, N);
    if (backward) {   // per-block partial sums: K6's persistent grid, the shade kernel's one block per 4 regions, or k_path's blocks
        const size_t shade_blocks = (max_regions + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE);
        size_t blocks = shade_blocks > (size_t)bwd_grid ? shade_blocks : (size_t)bwd_grid;
        const size_t path_blocks = (path_waves + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE);
        if (use_path && path_blocks > blocks)
            blocks = path_blocks;
        // rows per block: 24 for the register paths (<= 8 parameters), else one per parameter channel (LDS accumulators)
        const size_t rows = ctx->n_params <= DRT_FAST_PARAMS ? (size_t)DRT_FAST_PARAMS * 3
                                                             : (size_t)std::min(ctx->n_params, DRT_LDS_PARAMS) * 3;
        if ((rc = ensure(ctx, gpart_buf, blocks * rows * sizeof(double))) != DRT_OK) return rc;
    }

    BatchArgs a;
    memset(&a, 0, sizeof a);
    a.W = cam->width; a.H = cam->height; a.spp = spp;
    a.shard = rp->n_shards > 1 ? rp->shard : 0;
    a.n_shards = rp->n_shards > 1 ? rp->n_shards : 1;
    a.band = rp->band_rows > 0 ? rp->band_rows : 1;
    a.min_bounces = rp->min_bounces;
    a.depth_cap = D;
    a.cap_is_roulette = (rp->absorb >= 1.0 && rp->min_bounces == D) ? 1 : 0;
    a.absorb = rp->absorb;
    a.seed = rp->seed;
    a.rng_stream = drt_rng_stream(rp->seed, 0u);
    for (int i = 0; i < 3; ++i) {
        a.eye[i] = cam->eye[i]; a.fwd[i] = cam->forward[i];
        a.right[i] = cam->right[i]; a.up[i] = cam->up[i];
    }
    a.region_size = region_size;
    a.bvh_refill = getenv("DRT_HIP_BVH_REFILL") ? (uint32_t)atoi(getenv("DRT_HIP_BVH_REFILL")) : DRT_BVH_REFILL;
    a.bvh_descend_min = getenv("DRT_HIP_BVH_DESCEND_MIN") ? (uint32_t)atoi(getenv("DRT_HIP_BVH_DESCEND_MIN")) : DRT_BVH_DESCEND_MIN;
    a.region_shift = region_shift;
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

    R4* ra[2] = {(R4*)ctx->ray_a[0].p, (R4*)ctx->ray_a[1].p};
    typedef typename Q2<R>::T R2;
    R2* rb[2] = {(R2*)ctx->ray_b[0].p, (R2*)ctx->ray_b[1].p};
    uint2* rid[2] = {(uint2*)ctx->ray_id[0].p, (uint2*)ctx->ray_id[1].p};
    HitRec<R>* hit = (HitRec<R>*)ctx->hit.p;
    R4* lacc = (R4*)ctx->lacc.p;
    TapeRec<R>* tape = (TapeRec<R>*)ctx->tape.p;
    uint32_t* nv = (uint32_t*)ctx->nv.p;
    double* grad = (double*)ctx->grad[ctx->slot].p;
    double* gpart = (double*)gpart_buf.p;
    const int n_fast = ctx->n_params < DRT_FAST_PARAMS ? ctx->n_params : DRT_FAST_PARAMS;
    // gradient partials: gpart[block][g_stride], rows [0, g_rows) are reduced over the blocks by K7
    const bool g_general = ctx->n_params > DRT_FAST_PARAMS;
    const int g_rows = g_general ? std::min(ctx->n_params, DRT_LDS_PARAMS) * 3 : n_fast * 3;
    const int g_stride = g_general ? g_rows : DRT_FAST_PARAMS * 3;

    // Bounces per fused launch (at most 8).  Inside a launch the lanes of ended paths idle -- cheap next
    // to the queue traffic saved, measured: even at absorb = 0.5 four bounces per launch beat one --
    // so a launch only stops where fewer than ~10 % of its rays are expected to be left: ~7 % end
    // per bounce on a miss or a light (Cornell-like scenes), the roulette removes `absorb` of them at
    // every depth >= min_bounces.
    // DRT_HIP_SHADE_BOUNCES=n forces n (1 = one launch per bounce).
    auto bounces_from = [&](int k) -> int {
        if (!can_fuse)
            return 1;
        const int left = D - k;
        const int forced = shade_nb_env > 0 ? shade_nb_env : (rp->bounces_per_launch > 8 ? 8 : rp->bounces_per_launch);
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
    };
    uint64_t batch = 0;
    bool path_finished = false;
    for (uint32_t p0 = 0; p0 < n_local_pixels; p0 += Pb) {
        for (uint32_t s0 = 0; s0 < (uint32_t)spp; s0 += Sb, ++batch) {
            a.p0 = p0; a.s0 = s0;
            a.Pb = (n_local_pixels - p0) < Pb ? (n_local_pixels - p0) : Pb;
            a.Sb = ((uint32_t)spp - s0) < Sb ? ((uint32_t)spp - s0) : Sb;
            a.n_paths = a.Pb * a.Sb;
            a.n_regions = (a.n_paths + region_size - 1) / region_size;
            uint32_t* counts = (uint32_t*)counts_buf.p;   // reused by every batch (stream order)
            if (use_path) {
                // ---- the whole batch in ONE launch: camera -> path -> radiance sums + gradient partials
                PathArgs pa;
                memset(&pa, 0, sizeof pa);
                pa.W = a.W; pa.H = a.H; pa.spp = a.spp;
                pa.shard = a.shard; pa.n_shards = a.n_shards; pa.band = a.band;
                pa.Pb = a.Pb; pa.p0 = a.p0; pa.Sb = a.Sb; pa.s0 = a.s0;
                pa.spr = path_spr < a.Sb ? path_spr : a.Sb;
                pa.n_ranges = (a.Sb + pa.spr - 1) / pa.spr;
                pa.n_groups = (a.Pb + DRT_WAVE - 1) / DRT_WAVE;
                pa.min_bounces = a.min_bounces; pa.depth_cap = a.depth_cap; pa.cap_is_roulette = a.cap_is_roulette;
                pa.rr_threshold = a.rr_threshold; pa.seed = a.seed; pa.rng_stream = a.rng_stream;
                static const int regen_min_env = getenv("DRT_HIP_PATH_REGEN_MIN") ? atoi(getenv("DRT_HIP_PATH_REGEN_MIN")) : 8;
                pa.regen_min = (uint32_t)(regen_min_env < 1 ? 1 : regen_min_env);
                pa.p_rr = 1.0 - rp->absorb;
                pa.inv_p_rr = rp->absorb < 1.0 ? 1.0 / (1.0 - rp->absorb) : 0.0;   // (never used when every path ends at min_bounces)
                for (int i = 0; i < 3; ++i) {
                    pa.eye[i] = a.eye[i]; pa.fwd[i] = a.fwd[i]; pa.right[i] = a.right[i]; pa.up[i] = a.up[i];
                }
                pa.tan_half = a.tan_half; pa.aspect = a.aspect;
                pa.inv_W = 1.0 / (double)a.W; pa.inv_H = 1.0 / (double)a.H;
                const size_t n_waves = (size_t)pa.n_groups * pa.n_ranges;
                const int gpath = (int)((n_waves + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE));
                double* fpart = film ? (double*)fpart_buf.p : (double*)nullptr;
                double* gpix = gimg_param >= 0 ? (double*)ctx->gpix.p : (double*)nullptr;   // gradient image partials
                pa.gimg_param = gimg_param;
                // the kinds of the reference's own scene are compiled in (no per-shape branches); any other
                // analytic scene reads its kinds from the program
                static const bool sig_env = !(getenv("DRT_HIP_PATH_SIG") && atoi(getenv("DRT_HIP_PATH_SIG")) == 0);
                // (f64 too: the verification mode runs the same program with full-precision reciprocals and square roots)
                static const bool sig64_env = !(getenv("DRT_HIP_PATH_SIG_F64") && atoi(getenv("DRT_HIP_PATH_SIG_F64")) == 0);
                // (DRT_HIP_BUILTIN_PROGRAM=0: the reference's scene is specialised at run time like any other -- a test that the
                //  library's own build and hiprtc's agree bit for bit)
                static const bool builtin_env = !(getenv("DRT_HIP_BUILTIN_PROGRAM") && atoi(getenv("DRT_HIP_BUILTIN_PROGRAM")) == 0);
                const bool cornell_sig = sig_env && builtin_env && ctx->jit_mode >= 0 && (sizeof(R) == 4 || sig64_env) && ctx->n_shapes == DRT_NSIG_CORNELL && ctx->prog_sig[0] == DRT_SIG_CORNELL;
                unsigned long long* ptotal = path_finish ? (unsigned long long*)ctx->segtotal[ctx->slot].p : (unsigned long long*)nullptr;
                // (frames that overlap: this frame's grid goes to the slot's own stream, behind whoever still uses the slot's
                //  buffers, and the finishing launch on the context's stream waits for it.  Scene and parameter uploads block
                //  until they are done, so the grid needs nothing from the context's stream -- unless the call brings an adjoint
                //  image, which the caller may have produced in that stream's order: then the frame keeps its place in it.)
                hipStream_t ks = ctx->stream;
                const bool overlap = overlap_ok && path_finish;
                const int lane2 = ctx->slot & 1;                // which of the two k_path streams / sets of partial sums
                if (overlap) {
                    ks = ctx->path_stream[lane2];
                    if (d_adjoint) {
                        HIPCHK(ctx, hipEventRecord(ctx->ev_begin[lane2], ctx->stream));
                        HIPCHK(ctx, hipStreamWaitEvent(ks, ctx->ev_begin[lane2], 0));
                    }
                    // (the lane's buffers: their last user -- this lane's previous frame, or a render that went through the
                    //  context's stream -- has enqueued its last reader on the context's stream by the time its event is recorded)
                    if (ctx->lane_used[lane2] && ctx->ev_lane_free[lane2])
                        HIPCHK(ctx, hipStreamWaitEvent(ks, ctx->ev_lane_free[lane2], 0));
                }
                // tangents are carried for the parameters that ARE some BxDF's colour: 3 when the 4th is emission-only
                const bool three = ctx->max_colour_param < 3;
                const bool tangents = backward || gimg_param >= 0;
                // ---- a program of the scene's own (drt_jit.h): the instantiation for its KindSig, once it pays
                hipFunction_t jit = nullptr;
                ctx->scene_work += (uint64_t)a.n_paths * (uint64_t)(D > 0 ? D : 1);
                // (f32 only: the f64 verification mode keeps the reference's literal shape loop for every scene but the reference's own)
                if (!cornell_sig && sig_env && ctx->jit_mode > 0 && sizeof(R) == 4 &&
                    (ctx->jit_mode > 1 || ctx->scene_work >= DRT_JIT_AFTER_WORK)) {
                    const std::string sg = drt_jit::sig_type(ctx->prog_sig, ctx->n_shapes);
                    const char* rt = sizeof(R) == 4 ? "float" : "double";
                    const char* sp = ctx->has_specular ? "true" : "false";
                    char name[384];
                    if (unbiased)
                        snprintf(name, sizeof name, "k_path_unbiased<%s, %s, %d, %s>", rt, sp, ctx->n_params > 4 ? 8 : 4, sg.c_str());
                    else {
                        const int np = tangents ? (ctx->n_params > 4 ? 8 : 4) : 0;
                        const int nc = tangents ? (ctx->n_params > 4 ? 8 : (three ? 3 : 4)) : 0;
                        snprintf(name, sizeof name, "k_path<%s, %s, %d, %d, %s, %s>", rt, sp, np, nc, sg.c_str(), path_regen ? "true" : "false");
                    }
                    jit = jit_function(ctx, name);
                }
                st->path_program = cornell_sig ? DRT_PROGRAM_BUILTIN : (jit ? DRT_PROGRAM_SPECIALISED : DRT_PROGRAM_SORTED);
                if ((rc = timing_begin(ctx, timing, DRT_K_PATH)) != DRT_OK) return rc;
#define DRT_LAUNCH_PATH(SPEC, NP, NC, SG)                                                                                 \
    do {                                                                                                                 \
        if (path_regen)                                                                                                  \
            hipLaunchKernelGGL((k_path<R, SPEC, NP, NC, SG, true>), dim3(gpath), dim3(DRT_BLOCK), 0, ks,                 \
                               pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal, gpix);                   \
        else                                                                                                             \
            hipLaunchKernelGGL((k_path<R, SPEC, NP, NC, SG, false>), dim3(gpath), dim3(DRT_BLOCK), 0, ks,                \
                               pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal, gpix);                   \
    } while (0)
#define DRT_LAUNCH_PATH_SIG(SPEC, NP, NC)                                                  \
    do {                                                                                   \
        if (cornell_sig) DRT_LAUNCH_PATH(SPEC, NP, NC, SigCornell);                        \
        else DRT_LAUNCH_PATH(SPEC, NP, NC, SigNone);                                       \
    } while (0)
                if (jit) {
                    const DevScene<R>* a_scene = d_scene;
                    const R* a_params = d_params;
                    const float* a_adjoint = d_adjoint;
                    void* args_path[] = {&pa, &a_scene, &a_params, &a_adjoint, &gpart, &fpart, &counts, &ptotal, &gpix};
                    void* args_unb[] = {&pa, &a_scene, &a_params, &a_adjoint, &gpart, &fpart, &counts, &ptotal};
                    HIPCHK(ctx, hipModuleLaunchKernel(jit, (unsigned)gpath, 1, 1, DRT_BLOCK, 1, 1, 0, ks, unbiased ? args_unb : args_path, nullptr));
                } else if (unbiased) {                      // the unbiased operator: fresh suffix paths per vertex, in registers
#define DRT_LAUNCH_UNB(SPEC, NP)                                                                                              \
    do {                                                                                                                      \
        if (cornell_sig)                                                                                                      \
            hipLaunchKernelGGL((k_path_unbiased<R, SPEC, NP, SigCornell>), dim3(gpath), dim3(DRT_BLOCK), 0,                    \
                               ks, pa, d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal);                            \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_path_unbiased<R, SPEC, NP, SigNone>), dim3(gpath), dim3(DRT_BLOCK), 0, ks, pa,               \
                               d_scene, d_params, d_adjoint, gpart, fpart, counts, ptotal);                                    \
    } while (0)
                    if (ctx->n_params > 4) { if (ctx->has_specular) DRT_LAUNCH_UNB(true, 8); else DRT_LAUNCH_UNB(false, 8); }
                    else { if (ctx->has_specular) DRT_LAUNCH_UNB(true, 4); else DRT_LAUNCH_UNB(false, 4); }
#undef DRT_LAUNCH_UNB
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
#undef DRT_LAUNCH_PATH_SIG
#undef DRT_LAUNCH_PATH
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                if (overlap) {
                    HIPCHK(ctx, hipEventRecord(ctx->ev_path[lane2], ks));
                    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_path[lane2], 0));
                }
                st->launches[DRT_K_PATH]++;
                st->path_bytes += (film ? (uint64_t)pa.n_ranges * a.Pb * 3 * sizeof(double) : 0) +
                                  (backward ? (uint64_t)gpath * DRT_FAST_PARAMS * 3 * sizeof(double) : 0) + 2 * n_waves * sizeof(uint32_t);
                if (path_finish) {
                    // image, gradients and totals of the frame in one launch (timed in the film slot)
                    const uint32_t film_blocks = film ? (uint32_t)grid_for(ctx, a.Pb) : 0u;
                    const uint32_t grad_words = backward ? (uint32_t)ctx->n_params * 3u : 0u;
                    const uint32_t count_blocks = (uint32_t)std::min<size_t>(64, (n_waves + DRT_BLOCK - 1) / DRT_BLOCK);
                    if ((rc = timing_begin(ctx, timing, DRT_K_FILM)) != DRT_OK) return rc;
                    hipLaunchKernelGGL(k_path_finish, dim3(film_blocks + grad_words + count_blocks), dim3(DRT_BLOCK), 0, ctx->stream, pa,
                                       (const double*)fpart, d_out_rgb, film_blocks, (const double*)gpart, gpath, n_fast * 3,
                                       DRT_FAST_PARAMS * 3, grad, grad_words, (const uint32_t*)counts, (uint32_t)n_waves,
                                       (unsigned long long*)ctx->segtotal[ctx->slot].p);
                    if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                    st->launches[DRT_K_FILM]++;
                    st->units[DRT_K_FILM] += a.n_paths;
                    if (gpix && d_out_gimg) {   // the gradient image: the same sums over the sample ranges, its own output
                        const uint32_t gb = (uint32_t)grid_for(ctx, a.Pb);
                        hipLaunchKernelGGL(k_path_finish, dim3(gb), dim3(DRT_BLOCK), 0, ctx->stream, pa, (const double*)gpix, d_out_gimg, gb,
                                           (const double*)nullptr, 0, 0, DRT_FAST_PARAMS * 3, (double*)nullptr, 0u, (const uint32_t*)counts,
                                           0u, (unsigned long long*)ctx->segtotal[ctx->slot].p);
                    }
                    path_finished = true;
                    continue;
                }
                hipLaunchKernelGGL(k_sum_counts, dim3(64), dim3(DRT_BLOCK), 0, ctx->stream, counts, (uint32_t)(2 * n_waves),
                                   (unsigned long long*)ctx->segtotal[ctx->slot].p, (uint32_t)n_waves, 0ull, 0ull, 1u);
                if (backward) {
                    if ((rc = timing_begin(ctx, timing, DRT_K_GRADREDUCE)) != DRT_OK) return rc;
                    hipLaunchKernelGGL(k_gradreduce, dim3(n_fast > 0 ? n_fast * 3 : 1), dim3(DRT_BLOCK), 0, ctx->stream, gpart,
                                       gpath, n_fast * 3, grad, DRT_FAST_PARAMS * 3);
                    if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                    st->launches[DRT_K_GRADREDUCE]++;
                    st->units[DRT_K_GRADREDUCE] += (uint64_t)gpath;
                }
                if (film) {
                    if ((rc = timing_begin(ctx, timing, DRT_K_FILM)) != DRT_OK) return rc;
                    hipLaunchKernelGGL(k_film_parts, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, fpart,
                                       pa.n_ranges, a.Pb, a.p0, film);
                    if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                    st->launches[DRT_K_FILM]++;
                    st->units[DRT_K_FILM] += a.n_paths;
                }
                if (gpix && gfilm)
                    hipLaunchKernelGGL(k_film_parts, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, gpix,
                                       pa.n_ranges, a.Pb, a.p0, gfilm);
                continue;
            }
            HIPCHK(ctx, hipMemsetAsync(counts, 0, cw * sizeof(uint32_t), ctx->stream));
            const int g = (int)((a.n_regions + DRT_BLOCK / DRT_WAVE - 1) / (DRT_BLOCK / DRT_WAVE));
            const int gp = grid_for(ctx, a.n_paths);   // per-path kernels (K6): persistent grid
            const int gk2 = ctx->has_mesh ? k2_mesh_grid(ctx, a.n_paths) : gp;   // with a mesh every K2 wave leaves one candidate list
            const uint32_t k2w = (uint32_t)gk2 * (DRT_BLOCK / DRT_WAVE);
            const uint32_t cand_cap = (((a.n_regions << (region_shift - 6)) + k2w - 1) / k2w) * DRT_WAVE;

            // K1 folded into the first shade launch when it is a fused one and every path is alive at depth 0
            static const bool cam_env = !(getenv("DRT_HIP_FUSE_CAMERA") && atoi(getenv("DRT_HIP_FUSE_CAMERA")) == 0);
            // (only when that launch carries its rays through several bounces: with one launch per bounce the
            // depth-0 launch is the largest, and the camera's f64 math no longer hides behind K1's own writes)
            const bool camera_fused = cam_env && can_fuse && D > 0 && a.min_bounces > 0 && bounces_from(0) > 1;
            if (!camera_fused) {
                if ((rc = timing_begin(ctx, timing, DRT_K_RAYGEN)) != DRT_OK) return rc;
                // (scenes with a mesh: K1 also intersects its rays with the analytic shapes and builds the BVH walk's candidate
                //  lists -- hit lane `hit`, the one the bounce loop starts on; k_intersect is not launched at all)
                if (shade_tail) {
                    HIPCHK(ctx, hipMemsetAsync(ctx->cand_count.p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
                    hipLaunchKernelGGL((k_raygen<R, true>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene, ra[0], rb[0], rid[0], nv,
                                       counts, bvh, hit, (uint32_t*)ctx->cand.p, (R4*)ctx->cand_a.p, (R4*)ctx->cand_b.p,
                                       (uint32_t*)ctx->cand_count.p);
                } else
                    hipLaunchKernelGGL((k_raygen<R, false>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene, ra[0], rb[0], rid[0], nv,
                                       counts, bvh, (HitRec<R>*)nullptr, (uint32_t*)nullptr, (R4*)nullptr, (R4*)nullptr, (uint32_t*)nullptr);
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_RAYGEN]++;
                st->units[DRT_K_RAYGEN] += a.n_paths;
            }

            unsigned long long read_rows = 0, written_rows = 0;   // queue rows shade launches start from / end on
            for (int k = 0, lc = 0, nbk = 1, next_poll = DRT_POLL_EVERY; k < D; k += nbk, ++lc) {
                const int cur = lc & 1, nxt = cur ^ 1;
                nbk = bounces_from(k);
                if (!(camera_fused && k == 0)) read_rows |= 1ull << k;      // the camera launch generates its rays
                if (k + nbk < D) written_rows |= 1ull << (k + nbk);
                if (D > 2 * DRT_POLL_EVERY && k >= next_poll) {
                    next_poll = k + DRT_POLL_EVERY;
                    unsigned long long live = 0;
                    if ((rc = queue_length(ctx, counts + (size_t)k * max_regions, max_regions, &live)) != DRT_OK) return rc;
                    if (live == 0)
                        break;        // every path has ended: deeper queues stay empty
                }
                // unbiased: the camera ray's hit is the first chain vertex of the backward pass; K3 saves
                // it itself unless a mesh keeps the hit in a separate kernel's hands
                const bool fused = can_fuse;
                const bool save_here = unbiased && k == 0;
                // (fused or not, the shade launch has the ray and its final hit in registers)
                R4* sv_a = save_here ? cs.cv_a : (R4*)nullptr;
                typename Q2<R>::T* sv_b = save_here ? cs.cv_b : (typename Q2<R>::T*)nullptr;
                HitRec<R>* sv_hit = save_here ? cs.cv_hit : (HitRec<R>*)nullptr;
                // hit lane of this depth (double-buffered when the shade launch fills the next depth's itself)
                HitRec<R>* hit_k = shade_tail && (lc & 1) ? (HitRec<R>*)ctx->hit2.p : hit;
                HitRec<R>* hit_n = shade_tail ? ((lc & 1) ? hit : (HitRec<R>*)ctx->hit2.p) : (HitRec<R>*)nullptr;
                const bool lists_from_shade = shade_tail;               // (depth 0: from k_raygen<TAIL>)
                if (!fused) {
                    if (!lists_from_shade) {
                        if ((rc = timing_begin(ctx, timing, DRT_K_INTERSECT)) != DRT_OK) return rc;
                        hipLaunchKernelGGL(k_intersect<R>, dim3(gk2), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene,
                                           ra[cur], rb[cur], hit_k, counts + (size_t)k * max_regions, bvh,
                                           ctx->has_mesh ? (uint32_t*)ctx->cand.p : (uint32_t*)nullptr, (R4*)ctx->cand_a.p, (R4*)ctx->cand_b.p,
                                           (uint32_t*)ctx->cand_count.p, cand_cap, (unsigned long long*)ctx->segtotal[ctx->slot].p);
                        if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                        st->launches[DRT_K_INTERSECT]++;
                    }
                    if (ctx->has_mesh) {   // continues from the analytic hit: (t, primitive) refined by the BVH walk
                        if ((rc = timing_begin(ctx, timing, DRT_K_INTERSECT_MESH)) != DRT_OK) return rc;
                        const int gm = (int)std::min<uint64_t>(((uint64_t)a.n_paths + DRT_BLOCK - 1) / DRT_BLOCK, (uint64_t)ctx->n_cu * ctx->mesh_blocks_per_cu);
                        // lists: one per k_intersect wave, or one per queue region (half as long: handed out four at a time)
                        const uint32_t walk_lists = lists_from_shade ? a.n_regions : (uint32_t)gk2 * (DRT_BLOCK / DRT_WAVE);
                        const uint32_t walk_group = lists_from_shade ? shade_list_group : 1u;
                        hipLaunchKernelGGL(k_intersect_mesh<R>, dim3(gm), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene,
                                           bvh, hit_k, (const uint32_t*)ctx->cand.p, (const R4*)ctx->cand_a.p, (const R4*)ctx->cand_b.p,
                                           (uint32_t*)ctx->cand_count.p, lists_from_shade ? region_size : cand_cap,
                                           walk_lists, walk_group, coprime_multiplier((walk_lists + walk_group - 1) / walk_group),
                                           (unsigned long long*)ctx->segtotal[ctx->slot].p);
                        if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                        st->launches[DRT_K_INTERSECT_MESH]++;
                    }
                }
                TapeRec<R>* tape_k = tape + (size_t)k * a.n_paths;
                if ((rc = timing_begin(ctx, timing, DRT_K_SHADE)) != DRT_OK) return rc;
                {
                    int gs = g;                                // one wave per region, or persistent
                    static const int shade_bpc = getenv("DRT_HIP_SHADE_BLOCKS_PER_CU") ? atoi(getenv("DRT_HIP_SHADE_BLOCKS_PER_CU")) : 0;
                    if (shade_bpc > 0 && ctx->n_cu * shade_bpc < g)
                        gs = ctx->n_cu * shade_bpc;
                    uint32_t* ck = counts + (size_t)k * max_regions;
#define DRT_SHADE_NO_TAIL bvh, (HitRec<R>*)nullptr, (uint32_t*)nullptr, (R4*)nullptr, (R4*)nullptr, (uint32_t*)nullptr
#define DRT_LAUNCH_SHADE(SPEC, FUSE, SEG, DBASE)                                                               \
    hipLaunchKernelGGL((k_shade<R, SPEC, FUSE>), dim3(gs), dim3(DRT_BLOCK), 0, ctx->stream, a, k, nbk, d_scene, \
                       d_params, ra[cur], rb[cur], rid[cur], hit_k, ra[nxt], rb[nxt], rid[nxt], tape_k, nv,    \
                       ck, (uint32_t)max_regions, bvh.tri_shade, SEG, DBASE, sv_a, sv_b, sv_hit, DRT_SHADE_NO_TAIL)
#define DRT_LAUNCH_SHADE_TAIL(SPEC, SEG, DBASE)                                                                        \
    hipLaunchKernelGGL((k_shade<R, SPEC, false, false, true>), dim3(gs), dim3(DRT_BLOCK), 0, ctx->stream, a, k, nbk,   \
                       d_scene, d_params, ra[cur], rb[cur], rid[cur], hit_k, ra[nxt], rb[nxt], rid[nxt], tape_k, nv,   \
                       ck, (uint32_t)max_regions, bvh.tri_shade, SEG, DBASE, sv_a, sv_b, sv_hit, bvh,                  \
                       hit_n, (uint32_t*)ctx->cand.p, (R4*)ctx->cand_a.p, (R4*)ctx->cand_b.p, (uint32_t*)ctx->cand_count.p)
                    if (fused && camera_fused && k == 0) {
#define DRT_LAUNCH_CAMERA(SPEC)                                                                                       \
    hipLaunchKernelGGL((k_shade<R, SPEC, true, true>), dim3(gs), dim3(DRT_BLOCK), 0, ctx->stream, a, k, nbk,         \
                       d_scene, d_params, ra[cur], rb[cur], rid[cur], hit_k, ra[nxt], rb[nxt], rid[nxt], tape_k, nv, ck, \
                       (uint32_t)max_regions, bvh.tri_shade, 0, (const uint32_t*)nullptr, sv_a, sv_b, sv_hit, DRT_SHADE_NO_TAIL)
                        if (ctx->has_specular) DRT_LAUNCH_CAMERA(true);
                        else DRT_LAUNCH_CAMERA(false);
#undef DRT_LAUNCH_CAMERA
                    } else if (fused) {
                        if (ctx->has_specular) DRT_LAUNCH_SHADE(true, true, 0, (const uint32_t*)nullptr);
                        else DRT_LAUNCH_SHADE(false, true, 0, (const uint32_t*)nullptr);
                    } else if (shade_tail && k + nbk < D) {
                        // (the region lists of regions no wave visits stay empty)
                        HIPCHK(ctx, hipMemsetAsync(ctx->cand_count.p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
                        if (ctx->has_specular) DRT_LAUNCH_SHADE_TAIL(true, 0, (const uint32_t*)nullptr);
                        else DRT_LAUNCH_SHADE_TAIL(false, 0, (const uint32_t*)nullptr);
                    } else {
                        if (ctx->has_specular) DRT_LAUNCH_SHADE(true, false, 0, (const uint32_t*)nullptr);
                        else DRT_LAUNCH_SHADE(false, false, 0, (const uint32_t*)nullptr);
                    }
                }
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_SHADE]++;
            }

            hipLaunchKernelGGL(k_sum_counts, dim3(64), dim3(DRT_BLOCK), 0, ctx->stream, counts,
                               (uint32_t)((size_t)(D + 1) * max_regions), (unsigned long long*)ctx->segtotal[ctx->slot].p,
                               (uint32_t)max_regions, read_rows, written_rows, (uint32_t)D);
            if (backward && D > 0 && gimg_param >= 0) {
                // gradient image: per-path gradient of one parameter, averaged per pixel by K5
                if ((rc = timing_begin(ctx, timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
                hipLaunchKernelGGL(k_backward_image<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene,
                                   d_params, tape, nv, d_adjoint, (uint32_t)gimg_param, (R4*)ctx->gpath.p,
                                   film ? lacc : (R4*)nullptr);
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_BACKWARD]++;
                hipLaunchKernelGGL(k_film<R>, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, a,
                                   (const R4*)ctx->gpath.p, gfilm);
            } else if (unbiased && D > 0) {
                // forward radiance from the tape, then the adjoint rounds (see drt_kernels.h)
                if ((rc = timing_begin(ctx, timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
                if (film)
                    hipLaunchKernelGGL(k_radiance<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene, d_params,
                                       tape, nv, lacc);
                hipLaunchKernelGGL(k_adj_init<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, tape, nv, d_adjoint, cs);
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_BACKWARD]++;
                for (int r = 0; r < D; ++r) {
                    const int s = r + 1;
                    HIPCHK(ctx, hipMemsetAsync(counts + (size_t)s * max_regions, 0,
                                               (size_t)(D + 1 - s) * max_regions * sizeof(uint32_t), ctx->stream));
                    // (scenes with a mesh: the kernel also intersects the rays it queues with the analytic shapes and builds the
                    //  BVH walk's candidate lists -- hit lane `hit`, the one the suffix loop starts on)
#define DRT_LAUNCH_ADJ_VERTEX(SPEC, TAILV)                                                                                  \
    hipLaunchKernelGGL((k_adj_vertex<R, SPEC, TAILV>), dim3(g), dim3(DRT_BLOCK), 0, ctx->stream, a, r, d_scene, d_params, cs, \
                       bvh.tri_shade, ra[s & 1], rb[s & 1], rid[s & 1], nv, counts + (size_t)s * max_regions, bvh, hit,      \
                       (uint32_t*)ctx->cand.p, (R4*)ctx->cand_a.p, (R4*)ctx->cand_b.p, (uint32_t*)ctx->cand_count.p)
                    // (timed with the backward pass: it re-samples the chain vertex and queues the suffix's first ray)
                    if ((rc = timing_begin(ctx, timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
                    if (shade_tail) {
                        HIPCHK(ctx, hipMemsetAsync(ctx->cand_count.p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
                        if (ctx->has_specular) DRT_LAUNCH_ADJ_VERTEX(true, true);
                        else DRT_LAUNCH_ADJ_VERTEX(false, true);
                    } else {
                        if (ctx->has_specular) DRT_LAUNCH_ADJ_VERTEX(true, false);
                        else DRT_LAUNCH_ADJ_VERTEX(false, false);
                    }
                    if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                    st->launches[DRT_K_BACKWARD]++;
#undef DRT_LAUNCH_ADJ_VERTEX
                    bool chains_done = false;
                    if (D > 2 * DRT_POLL_EVERY && r >= 2) {
                        // no suffix ray queued in this round => every chain ends with this round
                        unsigned long long live = 0;
                        if ((rc = queue_length(ctx, counts + (size_t)s * max_regions, max_regions, &live)) != DRT_OK) return rc;
                        chains_done = live == 0;
                    }
                    unsigned long long sfx_read = 0, sfx_written = 0;     // rows relative to depth s
                    for (int k = s, lc = 0, nbk = 1, next_poll = s + DRT_POLL_EVERY; k < D && !chains_done; k += nbk, ++lc) {
                        const int cur = (s + lc) & 1, nxt = cur ^ 1;     // k_adj_vertex queued the suffix rays in buffer s & 1
                        nbk = bounces_from(k);
                        sfx_read |= 1ull << (k - s);
                        if (k + nbk < D) sfx_written |= 1ull << (k + nbk - s);
                        if (D > 2 * DRT_POLL_EVERY && k >= next_poll) {
                            next_poll = k + DRT_POLL_EVERY;
                            unsigned long long live = 0;
                            if ((rc = queue_length(ctx, counts + (size_t)k * max_regions, max_regions, &live)) != DRT_OK) return rc;
                            if (live == 0)
                                break;
                        }
                        uint32_t* ck = counts + (size_t)k * max_regions;
                        const bool fused = can_fuse;
                        // scenes with a mesh: every depth gets its analytic hit and its candidate lists from the launch that
                        // PRODUCES its rays (k_adj_vertex<TAIL> for depth s, k_shade<TAIL> after it), the hit lane double-buffered;
                        // the suffix's first ray and its FINAL hit -- the next chain vertex -- are saved by the shade launch of
                        // depth s, which holds both
                        HitRec<R>* hit_k = shade_tail && (lc & 1) ? (HitRec<R>*)ctx->hit2.p : hit;
                        HitRec<R>* hit_n = shade_tail ? ((lc & 1) ? hit : (HitRec<R>*)ctx->hit2.p) : (HitRec<R>*)nullptr;
                        const bool lists_from_shade = shade_tail;         // (depth s: from k_adj_vertex<TAIL>)
                        R4* sv_a = k == s ? cs.nx_a : (R4*)nullptr;
                        typename Q2<R>::T* sv_b = k == s ? cs.nx_b : (typename Q2<R>::T*)nullptr;
                        HitRec<R>* sv_hit = k == s ? cs.nx_hit : (HitRec<R>*)nullptr;
                        if (!fused) {
                            if (!lists_from_shade) {
                                if ((rc = timing_begin(ctx, timing, DRT_K_INTERSECT)) != DRT_OK) return rc;
                                hipLaunchKernelGGL(k_intersect<R>, dim3(gk2), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene,
                                                   ra[cur], rb[cur], hit_k, ck, bvh,
                                                   ctx->has_mesh ? (uint32_t*)ctx->cand.p : (uint32_t*)nullptr, (R4*)ctx->cand_a.p, (R4*)ctx->cand_b.p,
                                                   (uint32_t*)ctx->cand_count.p, cand_cap, (unsigned long long*)ctx->segtotal[ctx->slot].p);
                                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                                st->launches[DRT_K_INTERSECT]++;
                            }
                            if (ctx->has_mesh) {
                                if ((rc = timing_begin(ctx, timing, DRT_K_INTERSECT_MESH)) != DRT_OK) return rc;
                                const uint32_t walk_lists = lists_from_shade ? a.n_regions : (uint32_t)gk2 * (DRT_BLOCK / DRT_WAVE);
                                const uint32_t walk_group = lists_from_shade ? shade_list_group : 1u;
                                hipLaunchKernelGGL(k_intersect_mesh<R>, dim3((int)std::min<uint64_t>(((uint64_t)a.n_paths + DRT_BLOCK - 1) / DRT_BLOCK, (uint64_t)ctx->n_cu * ctx->mesh_blocks_per_cu)), dim3(DRT_BLOCK), 0,
                                                   ctx->stream, a, d_scene, bvh, hit_k, (const uint32_t*)ctx->cand.p,
                                                   (const R4*)ctx->cand_a.p, (const R4*)ctx->cand_b.p, (uint32_t*)ctx->cand_count.p,
                                                   lists_from_shade ? region_size : cand_cap, walk_lists, walk_group,
                                                   coprime_multiplier((walk_lists + walk_group - 1) / walk_group), (unsigned long long*)ctx->segtotal[ctx->slot].p);
                                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                                st->launches[DRT_K_INTERSECT_MESH]++;
                            }
                        }
                        TapeRec<R>* tape_k = tape + (size_t)k * a.n_paths;
                        const int gs = g;
                        if ((rc = timing_begin(ctx, timing, DRT_K_SHADE)) != DRT_OK) return rc;
                        if (fused) {
                            if (ctx->has_specular) DRT_LAUNCH_SHADE(true, true, s, (const uint32_t*)cs.dbase);
                            else DRT_LAUNCH_SHADE(false, true, s, (const uint32_t*)cs.dbase);
                        } else if (shade_tail && k + nbk < D) {
                            HIPCHK(ctx, hipMemsetAsync(ctx->cand_count.p, 0, (size_t)a.n_regions * sizeof(uint32_t), ctx->stream));
                            if (ctx->has_specular) DRT_LAUNCH_SHADE_TAIL(true, s, (const uint32_t*)cs.dbase);
                            else DRT_LAUNCH_SHADE_TAIL(false, s, (const uint32_t*)cs.dbase);
                        } else {
                            if (ctx->has_specular) DRT_LAUNCH_SHADE(true, false, s, (const uint32_t*)cs.dbase);
                            else DRT_LAUNCH_SHADE(false, false, s, (const uint32_t*)cs.dbase);
                        }
                        if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                        st->launches[DRT_K_SHADE]++;
                    }
                    if (s < D)
                        hipLaunchKernelGGL(k_sum_counts, dim3(64), dim3(DRT_BLOCK), 0, ctx->stream,
                                           counts + (size_t)s * max_regions, (uint32_t)((size_t)(D - s) * max_regions),
                                           (unsigned long long*)ctx->segtotal[ctx->slot].p, (uint32_t)max_regions, sfx_read, sfx_written, 0xFFFFFFFFu);
                    if ((rc = timing_begin(ctx, timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
                    if (ctx->n_params <= 4)
                        hipLaunchKernelGGL((k_adj_accumulate<R, 4>), dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, r,
                                           d_scene, d_params, tape, nv, cs, gpart, grad, g_rows, g_stride);
                    else if (ctx->n_params <= 8)
                        hipLaunchKernelGGL((k_adj_accumulate<R, 8>), dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, r,
                                           d_scene, d_params, tape, nv, cs, gpart, grad, g_rows, g_stride);
                    else
                        hipLaunchKernelGGL((k_adj_accumulate<R, 0>), dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, r,
                                           d_scene, d_params, tape, nv, cs, gpart, grad, g_rows, g_stride);
                    if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                    st->launches[DRT_K_BACKWARD]++;
                    std::swap(cs.cv_a, cs.nx_a);                // the suffix's first vertex is the chain's next one
                    std::swap(cs.cv_b, cs.nx_b);
                    std::swap(cs.cv_hit, cs.nx_hit);
                    if ((rc = timing_begin(ctx, timing, DRT_K_GRADREDUCE)) != DRT_OK) return rc;
                    hipLaunchKernelGGL(k_gradreduce, dim3(g_rows > 0 ? g_rows : 1), dim3(DRT_BLOCK), 0, ctx->stream, gpart, gp, g_rows, grad, g_stride);
                    if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                    st->launches[DRT_K_GRADREDUCE]++;
                    if (chains_done)
                        break;
                }
            } else if (backward && D > 0) {
                if ((rc = timing_begin(ctx, timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
#define DRT_LAUNCH_BWD(NP)                                                                              \
    hipLaunchKernelGGL((k_backward<R, NP>), dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene, d_params, \
                       tape, nv, d_adjoint, gpart, grad, film ? lacc : (R4*)nullptr, g_rows, g_stride)
                if (ctx->n_params <= 4) DRT_LAUNCH_BWD(4);
                else if (ctx->n_params <= 8) DRT_LAUNCH_BWD(8);
                else DRT_LAUNCH_BWD(0);
#undef DRT_LAUNCH_BWD
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_BACKWARD]++;
                if ((rc = timing_begin(ctx, timing, DRT_K_GRADREDUCE)) != DRT_OK) return rc;
                hipLaunchKernelGGL(k_gradreduce, dim3(g_rows > 0 ? g_rows : 1), dim3(DRT_BLOCK), 0, ctx->stream, gpart, gp, g_rows,
                                   grad, g_stride);
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_GRADREDUCE]++;
                st->units[DRT_K_GRADREDUCE] += (uint64_t)gp;
            }
            else if (D > 0 && film) {
                // forward only: radiance of every path from its tape
                if ((rc = timing_begin(ctx, timing, DRT_K_BACKWARD)) != DRT_OK) return rc;
                hipLaunchKernelGGL(k_radiance<R>, dim3(gp), dim3(DRT_BLOCK), 0, ctx->stream, a, d_scene, d_params,
                                   tape, nv, lacc);
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_BACKWARD]++;
            }
            if (D <= 0 && film)
                HIPCHK(ctx, hipMemsetAsync(lacc, 0, (size_t)a.n_paths * sizeof(R4), ctx->stream));
            if (film) {
                if ((rc = timing_begin(ctx, timing, DRT_K_FILM)) != DRT_OK) return rc;
                hipLaunchKernelGGL(k_film<R>, dim3(grid_for(ctx, a.Pb)), dim3(DRT_BLOCK), 0, ctx->stream, a,
                                   lacc, film);
                if ((rc = timing_end(ctx, timing)) != DRT_OK) return rc;
                st->launches[DRT_K_FILM]++;
                st->units[DRT_K_FILM] += a.n_paths;
            }
        }
    }
#undef DRT_LAUNCH_SHADE
    // debugging aid: DRT_HIP_DUMP_PATH=<path index in the last batch> prints that path's tape
    if (const char* e = getenv("DRT_HIP_DUMP_PATH")) {
        const size_t i = (size_t)atoll(e);
        if (i < a.n_paths && D > 0) {
            (void)hipStreamSynchronize(ctx->stream);
            uint32_t k_nv = 0;
            (void)hipMemcpy(&k_nv, nv + i, sizeof k_nv, hipMemcpyDeviceToHost);
            fprintf(stderr, "[drt_hip] path %zu: %u vertices\n", i, k_nv);
            for (uint32_t k = 0; k < k_nv && k < (uint32_t)D; ++k) {
                TapeRec<R> tr;
                (void)hipMemcpy(&tr, tape + (size_t)k * a.n_paths + i, sizeof tr, hipMemcpyDeviceToHost);
                fprintf(stderr, "[drt_hip]   k=%u m=%.9g colour=%u emission=%u\n", k, (double)tr.m, tr.ids & 0xFFFFu, tr.ids >> 16);
            }
        }
    }
    st->batches = batch;
    st->paths = total_paths;
    if (film && d_out_rgb && !path_finished) {
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, n_local_pixels)), dim3(DRT_BLOCK), 0, ctx->stream,
                           a, n_local_pixels, film, d_out_rgb);
    }
    if (gimg_param >= 0 && gfilm && d_out_gimg && !path_finished) {
        hipLaunchKernelGGL(k_resolve, dim3(grid_for(ctx, n_local_pixels)), dim3(DRT_BLOCK), 0, ctx->stream,
                           a, n_local_pixels, gfilm, d_out_gimg);
    }
    {   // this render's lane of partial-sum buffers is free once the context's stream has come this far
        const int lane = overlap_ok ? (ctx->slot & 1) : 0;
        if (ctx->ev_lane_free[lane]) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_lane_free[lane], ctx->stream));
            ctx->lane_used[lane] = true;
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return DRT_OK;
}

} // namespace

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
    if (const char* e = getenv("DRT_HIP_JIT"))
        ctx->jit_mode = !strcmp(e, "force") ? DRT_SPECIALISE_NOW : (atoi(e) > 0 ? DRT_SPECIALISE_AUTO : (atoi(e) < 0 ? DRT_SPECIALISE_GENERIC : DRT_SPECIALISE_NEVER));
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return DRT_ERR_HIP;
    }
    {   // Every stream of the context is made HERE, in this order, before the process makes any other: which of them run
        // side by side depends on the order HIP has seen them in (measured: the two k_path streams made later, next to the
        // copy stream, never overlapped their grids; made here they do -- and the copy stream made later, after them, no
        // longer overlapped its launch with the next frame: 0.81 -> 0.90 ms through host buffers).
        static const bool overlap_env = !(getenv("DRT_HIP_OVERLAP_FRAMES") && atoi(getenv("DRT_HIP_OVERLAP_FRAMES")) == 0);
        for (int i = 0; i < 2 && overlap_env; ++i)
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
    }
    {   // the BVH walk is a persistent kernel whose waves own strided streams of rays: its grid must be exactly what
        // is resident at once (more blocks would run as a second round behind the first, at half the occupancy)
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_intersect_mesh<float>, DRT_BLOCK, 0) == hipSuccess && nb > 0)
            ctx->mesh_blocks_per_cu = nb;
        (void)hipGetLastError();
        if (const char* e = getenv("DRT_HIP_MESH_BLOCKS_PER_CU"))
            if (atoi(e) > 0)
                ctx->mesh_blocks_per_cu = atoi(e);
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
    DevBuf* bufs[] = {&ctx->fpart2, &ctx->gpart2, &ctx->counts2, &ctx->fpart, &ctx->gpix, &ctx->cand, &ctx->cand_a, &ctx->cand_b, &ctx->cand_count, &ctx->ray_a[0], &ctx->ray_a[1], &ctx->ray_b[0], &ctx->ray_b[1], &ctx->ray_id[0], &ctx->ray_id[1], &ctx->hit, &ctx->hit2, &ctx->lacc, &ctx->gpath, &ctx->gfilm, &ctx->gimg_out, &ctx->tape, &ctx->nv,
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
    if (ctx->d_scene_f) (void)hipFree(ctx->d_scene_f);
    if (ctx->d_scene_d) (void)hipFree(ctx->d_scene_d);
    if (ctx->d_params_f) (void)hipFree(ctx->d_params_f);
    if (ctx->d_params_d) (void)hipFree(ctx->d_params_d);
    if (ctx->h_probe)
        (void)hipHostFree(ctx->h_probe);
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
    release(ctx->probe);
    for (hipEvent_t e : ctx->event_pool)
        (void)hipEventDestroy(e);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int upload_scene_one(drt_hip_ctx* ctx, const drt_scene_desc* s)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!s || s->n_shapes < 0 || s->n_shapes > DRT_MAX_SHAPES || s->n_materials < 0 ||
        s->n_materials > DRT_MAX_MATERIALS || s->n_emitters < 0 || s->n_emitters > DRT_MAX_EMITTERS ||
        s->n_params < 0 || s->n_params + 1 >= (int)DRT_ID_NONE ||
        (s->n_shapes && !s->shapes) || (s->n_materials && !s->materials) ||
        (s->n_emitters && !s->emitters) || (s->n_params && !s->params))
        return fail(ctx, DRT_ERR_INVALID, "scene: bad counts or null arrays");
    for (int i = 0; i < s->n_shapes; ++i) {
        const drt_shape_desc& sh = s->shapes[i];
        if (sh.type == DRT_SHAPE_MESH) {
            if (sh.mesh < 0 || sh.mesh >= s->n_meshes || !s->meshes)
                return fail(ctx, DRT_ERR_INVALID, "scene: mesh index out of range");
            const drt_mesh_desc& m = s->meshes[sh.mesh];
            if (m.n_triangles < 0 || m.n_vertices < 0 || (m.n_triangles && (!m.vertices || !m.indices)))
                return fail(ctx, DRT_ERR_INVALID, "scene: malformed mesh");
            for (int t = 0; t < m.n_triangles * 3; ++t)
                if (m.indices[t] >= (uint32_t)m.n_vertices)
                    return fail(ctx, DRT_ERR_INVALID, "scene: mesh vertex index out of range");
            if (m.face_material)
                for (int t = 0; t < m.n_triangles; ++t)
                    if (m.face_material[t] < -1 || m.face_material[t] >= s->n_materials)
                        return fail(ctx, DRT_ERR_INVALID, "scene: face material index out of range");
        } else if (sh.type != DRT_SHAPE_PLANE && sh.type != DRT_SHAPE_SPHERE)
            return fail(ctx, DRT_ERR_INVALID, "scene: unknown shape type");
        if (sh.material < -1 || sh.material >= s->n_materials || sh.emitter < -1 || sh.emitter >= s->n_emitters)
            return fail(ctx, DRT_ERR_INVALID, "scene: shape material/emitter index out of range");
    }
    for (int i = 0; i < s->n_materials; ++i) {
        if (s->materials[i].type == DRT_BXDF_MIRROR)
            continue;                  // no colour parameter
        if (s->materials[i].type != DRT_BXDF_DIFFUSE && s->materials[i].type != DRT_BXDF_SPECULAR)
            return fail(ctx, DRT_ERR_INVALID, "scene: unknown material type");
        if (s->materials[i].param < 0 || s->materials[i].param >= s->n_params)
            return fail(ctx, DRT_ERR_INVALID, "scene: material parameter index out of range");
    }
    for (int i = 0; i < s->n_emitters; ++i)
        if (s->emitters[i].param < 0 || s->emitters[i].param >= s->n_params)
            return fail(ctx, DRT_ERR_INVALID, "scene: emitter parameter index out of range");

    // (drt_hip_wait hands a frame over with the scene's parameter count and requires_grad flags: they must still be the ones
    //  the frame was rendered with)
    for (int i = 0; i < DRT_HIP_FRAMES_IN_FLIGHT; ++i)
        if (ctx->in_flight[i])
            return fail(ctx, DRT_ERR_INVALID, "upload_scene: asynchronous frames are in flight -- drt_hip_wait for them first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // from here on the device state is being replaced: a failure below (BVH limits, out of memory) must leave the
    // context WITHOUT a scene, not with the new records under the old scene's bookkeeping
    ctx->has_scene = false;
    DevScene<float>* hf = new DevScene<float>();
    DevScene<double>* hd = new DevScene<double>();
    std::vector<float> pf;
    std::vector<double> pd;
    unsigned long long sig[4];
    fill_scene(*hf, pf, s, sig);
    fill_scene(*hd, pd, s, sig);
    int rc = DRT_OK;
    auto up = [&](void** dst, const void* src, size_t bytes) -> int {
        if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
        hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
        if (e == hipSuccess && bytes)
            e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            ctx->err = std::string("scene upload: ") + hipGetErrorString(e);
            return DRT_ERR_HIP;
        }
        return DRT_OK;
    };
    if (rc == DRT_OK) rc = up((void**)&ctx->d_scene_f, hf, sizeof *hf);
    if (rc == DRT_OK) rc = up((void**)&ctx->d_scene_d, hd, sizeof *hd);
    if (rc == DRT_OK) rc = up((void**)&ctx->d_params_f, pf.data(), pf.size() * sizeof(float));
    if (rc == DRT_OK) rc = up((void**)&ctx->d_params_d, pd.data(), pd.size() * sizeof(double));
    const int n_dev_params = hf->n_params;   // user parameters + internal constants
    ctx->prog_ok = hf->prog_ok != 0;
    ctx->prog_sorted = hf->prog_sorted != 0;
    for (int i = 0; i < 4; ++i)
        ctx->prog_sig[i] = sig[i];
    ctx->max_colour_param = -1;
    for (int i = 0; i < hf->n_materials; ++i)
        ctx->max_colour_param = std::max(ctx->max_colour_param, hf->materials[i].param);
    delete hf;
    delete hd;
    if (rc != DRT_OK)
        return rc;
    release_mesh(ctx);
    {
        std::vector<drt_bvh::Tri> tris;
        uint32_t flat = 0;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < s->n_shapes; ++i) {
            const drt_shape_desc& sh = s->shapes[i];
            if (sh.type != DRT_SHAPE_MESH) { ++flat; continue; }
            const drt_mesh_desc& m = s->meshes[sh.mesh];
            for (int k = 0; k < m.n_triangles; ++k, ++flat) {
                drt_bvh::Tri t;
                const double* a = &m.vertices[(size_t)m.indices[k * 3] * 3];
                const double* b = &m.vertices[(size_t)m.indices[k * 3 + 1] * 3];
                const double* c = &m.vertices[(size_t)m.indices[k * 3 + 2] * 3];
                for (int x = 0; x < 3; ++x) {
                    t.v0[x] = a[x]; t.e1[x] = b[x] - a[x]; t.e2[x] = c[x] - a[x];
                    lo[x] = std::min(lo[x], std::min(a[x], std::min(b[x], c[x])));
                    hi[x] = std::max(hi[x], std::max(a[x], std::max(b[x], c[x])));
                }
                // normalize(cross(e1, e2)) in the operation order of the oracle / harness
                const double nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
                const double ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
                const double nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
                const double len = sqrt(((0.0 + nx * nx) + ny * ny) + nz * nz);
                t.n[0] = nx / len; t.n[1] = ny / len; t.n[2] = nz / len;
                t.global = (uint32_t)tris.size();
                t.flat = flat;
                const int mat = m.face_material ? m.face_material[k] : sh.material;
                t.ids = (uint32_t)(mat < 0 ? 0xFFFF : mat) | ((uint32_t)(sh.emitter < 0 ? 0xFFFF : sh.emitter) << 16);
                tris.push_back(t);
            }
        }
        if (!tris.empty()) {
            if (tris.size() >= (1u << 28))
                return fail(ctx, DRT_ERR_UNSUPPORTED, "scene: more than 2^28 triangles");
            const double diag = sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) +
                                     (hi[2] - lo[2]) * (hi[2] - lo[2]));
            static_assert(drt_bvh::kStackEntries == DRT_BVH_STACK, "builder and traversal kernel disagree on the stack size");
            // Padding of the boxes: 1e-5 of the mesh diagonal, and never less than 2e-6 of the scene's extent -- the f32
            // walk places a box plane to ~2^-22 of the distance between ray origin and node (k_intersect_mesh), and ray
            // origins lie on the scene's surfaces; a mesh that is tiny against its room keeps conservative boxes
            // (tools/tiny_mesh.py: no lost hit down to 1/256 of the config-4 mesh).
            double extent = 0;
            for (int x = 0; x < 3; ++x)
                extent = std::max(extent, std::max(fabs(lo[x]), fabs(hi[x])));
            for (int i = 0; i < s->n_shapes; ++i) {
                const drt_shape_desc& sh = s->shapes[i];
                if (sh.type == DRT_SHAPE_PLANE) {          // (the normal is not normalised, shape.hpp:58-59)
                    const double nn = sqrt(sh.p[0] * sh.p[0] + sh.p[1] * sh.p[1] + sh.p[2] * sh.p[2]);
                    if (nn > 0)
                        extent = std::max(extent, fabs(sh.p[3]) / nn);
                }
                else if (sh.type == DRT_SHAPE_SPHERE)
                    extent = std::max(extent, sqrt(sh.p[0] * sh.p[0] + sh.p[1] * sh.p[1] + sh.p[2] * sh.p[2]) + fabs(sh.p[3]));
            }
            const double pad = std::max(1e-5 * (diag > 0 ? diag : 1.0), 2e-6 * extent);
            const drt_bvh::Built built = drt_bvh::build(tris, DRT_BVH_LDS_NODES, pad);
            if (built.stack_need > DRT_BVH_STACK)      // not even a balanced tree fits (> ~2 M triangles)
                return fail(ctx, DRT_ERR_UNSUPPORTED, "scene: the BVH of this mesh needs a deeper traversal stack than the device kernel has");
            if ((rc = upload_bvh<float>(ctx, built, tris, &ctx->bvh_f)) != DRT_OK) return rc;
            if ((rc = upload_bvh<double>(ctx, built, tris, &ctx->bvh_d)) != DRT_OK) return rc;
            ctx->has_mesh = true;
            ctx->bvh_bytes = (uint64_t)built.nodes.size() * 64 + (uint64_t)tris.size() * 48;   // f32 image: nodes + three 16-byte triangle lanes
        }
    }
    ctx->scene_work = 0;
    ctx->n_user_params = s->n_params;
    ctx->n_params = n_dev_params;      // + the internal constant of mirror materials, if any
    ctx->n_shapes = s->n_shapes;
    ctx->requires_grad.assign((size_t)ctx->n_params, 1);
    if (ctx->n_params > s->n_params)
        ctx->requires_grad[(size_t)s->n_params] = 0;
    if (s->requires_grad)
        for (int i = 0; i < s->n_params; ++i)
            ctx->requires_grad[i] = s->requires_grad[i] ? 1 : 0;
    // only materials that a shape or a mesh face actually uses decide the K3 instantiation
    // (render.cpp:35 creates a specular material its scene never uses)
    ctx->has_specular = false;
    // (mirrors live in the specular instantiation too)
    auto uses = [&](int m) { if (m >= 0 && s->materials[m].type != DRT_BXDF_DIFFUSE) ctx->has_specular = true; };
    for (int i = 0; i < s->n_shapes; ++i) {
        uses(s->shapes[i].material);
        if (s->shapes[i].type == DRT_SHAPE_MESH && s->meshes[s->shapes[i].mesh].face_material)
            for (int t = 0; t < s->meshes[s->shapes[i].mesh].n_triangles; ++t)
                uses(s->meshes[s->shapes[i].mesh].face_material[t]);
    }
    ctx->has_scene = true;
    return DRT_OK;
}

static int update_params_one(drt_hip_ctx* ctx, const double* params)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!ctx->has_scene)
        return fail(ctx, DRT_ERR_NO_SCENE, "update_params before upload_scene");
    if (!params)
        return fail(ctx, DRT_ERR_INVALID, "params is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<float> pf((size_t)ctx->n_user_params * 3);   // internal constants keep their values
    for (size_t i = 0; i < pf.size(); ++i)
        pf[i] = (float)params[i];
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(ctx->d_params_f, pf.data(), pf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_params_d, params, pf.size() * sizeof(double), hipMemcpyHostToDevice));
    return DRT_OK;
}

} // extern "C"

// ---- one render call in phases (see RenderJob) ---------------------------------------------------

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
        if (j.dev_out) {
            j.d_out = out_rgb;
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
                if ((rc = ensure(ctx, ctx->adjoint, npix_all * 3 * sizeof(float))) != DRT_OK) return rc;
                HIPCHK(ctx, hipMemcpyAsync(ctx->adjoint.p, adjoint_rgb, npix_all * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
                d_adj = (const float*)ctx->adjoint.p;
            }
        }
    }
    if (gimg_param >= 0) {
        const size_t fb = (size_t)(j.n_local_pixels ? j.n_local_pixels : 1) * 3 * sizeof(double);
        if ((rc = ensure(ctx, ctx->gfilm, fb)) != DRT_OK) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->gfilm.p, 0, fb, ctx->stream));
        if (j.dev_out) {
            j.d_gimg = out_gimg;
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

// the context's pinned block of one render: [totals 64 B | gradients | image | gradient image]
static int ensure_stage(drt_hip_ctx* ctx, RenderJob& j)
{
    const size_t npix_all = (size_t)j.cam.width * j.cam.height;
    j.img_bytes = npix_all * 3 * sizeof(float);
    j.grad_bytes = (size_t)ctx->n_user_params * 3 * sizeof(double);
    j.off_grad = 64;
    j.off_img = j.off_grad + ((j.grad_bytes + 15) & ~(size_t)15);
    j.off_gimg = j.off_img + j.img_bytes;
    const size_t need = j.off_gimg + j.img_bytes;
    if (ctx->h_stage_cap[ctx->slot] < need) {
        if (ctx->h_stage[ctx->slot])
            (void)hipHostFree(ctx->h_stage[ctx->slot]);
        ctx->h_stage[ctx->slot] = nullptr;
        ctx->h_stage_cap[ctx->slot] = 0;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_stage[ctx->slot], need));
        ctx->h_stage_cap[ctx->slot] = need;
    }
    return DRT_OK;
}

// asynchronous host-buffer renders: gradients and totals of the frame -> the pinned block, written by the device (one
// small launch in stream order; the image got there from the finishing kernels)
__global__ void __launch_bounds__(DRT_WAVE) k_results_to_host(const double* __restrict__ grad, int n_grad, const uint8_t* __restrict__ requires_grad_dev,
                                                              const unsigned long long* __restrict__ totals, double* __restrict__ h_grad,
                                                              unsigned long long* __restrict__ h_totals)
{
    (void)requires_grad_dev;
    for (int i = threadIdx.x; i < n_grad; i += DRT_WAVE)
        h_grad[i] = grad[i];
    if (threadIdx.x < DRT_TOTAL_WORDS)
        h_totals[threadIdx.x] = totals[threadIdx.x];
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
    if (j.zero_copy) {
        // (asynchronous host-buffer render: the image is in the pinned block already -- the finishing kernels wrote it
        //  there; gradients and totals follow by one small launch)
        ctx->h_segments = 0;
        j.want_segments = j.stats && j.n_count_words;
        hipLaunchKernelGGL(k_results_to_host, dim3(1), dim3(DRT_WAVE), 0, cs, (const double*)ctx->grad[ctx->slot].p,
                           (j.backward && j.out_param_grad && with_grad) ? ctx->n_user_params * 3 : 0, (const uint8_t*)nullptr,
                           (const unsigned long long*)ctx->segtotal[ctx->slot].p, (double*)(ctx->h_stage[ctx->slot] + j.off_grad),
                           (unsigned long long*)ctx->h_stage[ctx->slot]);
        HIPCHK(ctx, hipGetLastError());
        return DRT_OK;
    }
    if (j.copy_kernel) {
        // (asynchronous host-buffer render, two-stream form: everything of the frame crosses the link in one launch on the
        //  copy stream while the next frame's kernels run)
        ctx->h_segments = 0;
        j.want_segments = j.stats && j.n_count_words;
        const bool img = j.out_rgb && j.n_local_pixels;
        static const int copy_blocks = getenv("DRT_HIP_COPY_BLOCKS") ? std::max(1, atoi(getenv("DRT_HIP_COPY_BLOCKS"))) : 64;
        hipLaunchKernelGGL(k_frame_to_host, dim3(copy_blocks), dim3(DRT_BLOCK), 0, cs, img ? (const float*)j.d_out : (const float*)nullptr,
                           (float*)(ctx->h_stage[ctx->slot] + j.off_img), (uint32_t)j.cam.width * 3u,
                           (uint32_t)(j.n_local_pixels / (uint32_t)j.cam.width), (uint32_t)j.band, (uint32_t)j.n_shards, (uint32_t)j.shard,
                           (const double*)ctx->grad[ctx->slot].p, (j.backward && j.out_param_grad && with_grad) ? ctx->n_user_params * 3 : 0,
                           (const unsigned long long*)ctx->segtotal[ctx->slot].p, (double*)(ctx->h_stage[ctx->slot] + j.off_grad),
                           (unsigned long long*)ctx->h_stage[ctx->slot]);
        HIPCHK(ctx, hipGetLastError());
        return DRT_OK;
    }
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
        if (j.out_rgb && j.n_local_pixels)
            rows_to_stage(j.d_out, j.off_img);
        if (j.gimg_param >= 0 && j.out_gimg && j.n_local_pixels)
            rows_to_stage(j.d_gimg, j.off_gimg);
        HIPCHK(ctx, e);
        if (j.backward && j.out_param_grad && with_grad)
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage[ctx->slot] + j.off_grad, ctx->grad[ctx->slot].p, j.grad_bytes, hipMemcpyDeviceToHost, cs));
    }
    ctx->h_segments = 0;
    j.want_segments = j.stats && j.n_count_words;
    if (j.want_segments)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage[ctx->slot], ctx->segtotal[ctx->slot].p, DRT_TOTAL_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
    return DRT_OK;
}

// phase 4: wait (unless the caller asked for an asynchronous device-pointer render), hand over, statistics
static int render_finish(drt_hip_ctx* ctx, bool with_grad = true, hipEvent_t done = nullptr)
{
    RenderJob& j = ctx->job;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (done)
        HIPCHK(ctx, hipEventSynchronize(done));       // (an asynchronous render: its copies are complete; later frames may still run)
    else if (j.sync)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
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
        if (j.out_rgb && j.n_local_pixels)
            rows_to_caller(j.out_rgb, j.off_img);
        if (j.gimg_param >= 0 && j.out_gimg && j.n_local_pixels)
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
        static const bool threads_env = !(getenv("DRT_HIP_GROUP_THREADS") && atoi(getenv("DRT_HIP_GROUP_THREADS")) == 0);
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
    rc = render_launch(ctx, cam, rp, adjoint_rgb, out_rgb, out_param_grad, stats, gimg_param, out_gimg);
    ctx->overlap_next = false;
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

extern "C" {

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
    static const bool two_streams = !(getenv("DRT_HIP_ASYNC_COPY") && !strcmp(getenv("DRT_HIP_ASYNC_COPY"), "inline"));
    ctx->zero_copy_next = !two_streams;
    // (the k_path grids of consecutive frames overlap -- render_impl: path_stream --: frame t shares its stream and its set of
    //  partial sums with frame t - 2, whose finishing launch, on the context's stream, must have read them)
    ctx->overlap_next = two_streams && !(rp && (rp->flags & DRT_RENDER_SERIAL));
    int rc = render_launch(ctx, cam, rp, adjoint_rgb, out_rgb, out_param_grad, &sink, -1, nullptr);
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
extern "C" int drt_hip_debug_bvh_stats(unsigned long long* out8)
{
    (void)hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_bvh_stats), 8 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    unsigned long long zero[8] = {0};
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
    const drt_jit::Code& c = drt_jit::compile(arch, name_expr);
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
