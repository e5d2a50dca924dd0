// drt_scene.h -- drt_hip_upload_scene / drt_hip_update_params: the caller's POD scene (include/drt_hip.h) -> the device's
// records (DevScene: shapes, materials, the kind-sorted intersection program and its signature), the BVH of its triangle
// meshes (drt_bvh.h) in both compute types, the parameter vectors.  Replaces the scene block of the reference's
// src/render.cpp:26-59.
#pragma once

namespace {

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
        ds.shapes[i].type = s->shapes[i].type == DRT_SHAPE_USER ? DRT_SHAPE_USER + s->shapes[i].mesh : s->shapes[i].type;
        if (s->shapes[i].type == DRT_SHAPE_USER)
            for (int j = 0; j < 4; ++j)
                ds.user_q[i][j] = s->user_params ? (R)s->user_params[(size_t)i * 4 + j] : R(0);
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
        if (s->materials[i].type >= DRT_BXDF_USER)      // a caller-defined kind: (exponent, norm) = the two values of its record
            ds.materials[i].norm = (s->n_kinds != 0 && s->user_bxdf_params) ? (R)s->user_bxdf_params[i] : R(0);
    }
    for (int i = 0; i < s->n_emitters; ++i)
        ds.emitter_param[i] = s->emitters[i].param;
    // rows of the one-launch kernels' gradient tables: the parameters that require a gradient, in order (the mirrors' constant never)
    for (int i = 0; i < DRT_PATH_LDS_PARAMS; ++i) {
        const bool wants = i < s->n_params && (!s->requires_grad || s->requires_grad[i]);
        ds.grad_slot[i] = wants ? (unsigned short)ds.n_grad_slots++ : (unsigned short)DRT_SLOT_NONE;
    }
    // k_path (drt_path.h): the parameter ids of every shape in one word, and the intersection program
    ds.prog_ok = 1;
    bool has_mesh_shape = false, has_user_shape = false;
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
        } else if (s->shapes[i].type == DRT_SHAPE_USER) {
            // a caller-defined kind: tested by the kernel hiprtc compiles for this scene (drt_prog.h), by no other program
            has_user_shape = true;
            kinds[i] = DRT_PK_USER0 + s->shapes[i].mesh;
            for (int j = 0; j < 4; ++j)
                recs[i][j] = (R)s->shapes[i].p[j];
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
        for (int k = 0; k < DRT_PK_USER0 + DRT_MAX_USER_KINDS; ++k) {     // (the user kinds' records too: KindSig::pos counts them)
            ds.kind_begin[k] = n;
            for (int i = 0; i < s->n_shapes; ++i)
                if (kinds[i] == k) {
                    for (int j = 0; j < 4; ++j)
                        ds.sorted[n][j] = recs[i][j];
                    ds.sorted_shape[n++] = i;
                }
        }
        for (int k = DRT_PK_USER0 + DRT_MAX_USER_KINDS; k < 8; ++k)
            ds.kind_begin[k] = n;
    }
    ds.prog_sorted = has_user_shape ? 0 : ds.prog_ok;   // the sorted program is valid (for the analytic shapes; it knows no caller-defined kind)
    if (has_mesh_shape)
        ds.prog_ok = 0;                         // ... but k_path is not for scenes with a mesh
    params.assign((size_t)ds.n_params * 3, R(1));
    for (size_t i = 0; i < (size_t)s->n_params * 3; ++i)
        params[i] = (R)s->params[i];
}

// event-bracketed launch bookkeeping (DRT_RENDER_TIMING)

int upload_scene_one(drt_hip_ctx* ctx, const drt_scene_desc* s)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!s || s->n_shapes < 0 || s->n_shapes > DRT_MAX_SHAPES || s->n_materials < 0 ||
        s->n_materials > DRT_MAX_MATERIALS || s->n_emitters < 0 || s->n_emitters > DRT_MAX_EMITTERS ||
        s->n_params < 0 || s->n_params + 1 >= (int)DRT_ID_NONE ||
        (s->n_shapes && !s->shapes) || (s->n_materials && !s->materials) ||
        (s->n_emitters && !s->emitters) || (s->n_params && !s->params))
        return fail(ctx, DRT_ERR_INVALID, "scene: bad counts or null arrays");
    bool any_user = false, any_mesh = false;
    for (int i = 0; i < s->n_shapes; ++i) {
        const drt_shape_desc& sh = s->shapes[i];
        if (sh.type == DRT_SHAPE_USER) {
            any_user = true;
            if (s->n_kinds < 1 || s->n_kinds > DRT_MAX_USER_KINDS || !s->kinds)
                return fail(ctx, DRT_ERR_INVALID, "scene: a DRT_SHAPE_USER shape needs 1 .. DRT_MAX_USER_KINDS entries in drt_scene_desc.kinds");
            if (sh.mesh < 0 || sh.mesh >= s->n_kinds)
                return fail(ctx, DRT_ERR_INVALID, "scene: shape kind index out of range");
            if (!s->kinds[sh.mesh].intersect_src || !s->kinds[sh.mesh].normal_src)
                return fail(ctx, DRT_ERR_INVALID, "scene: a shape kind without intersect / normal source");
        }
        any_mesh = any_mesh || sh.type == DRT_SHAPE_MESH;
        if (sh.type == DRT_SHAPE_USER) {
        } else if (sh.type == DRT_SHAPE_MESH) {
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
            if (m.face_param)
                for (int t = 0; t < m.n_triangles; ++t)
                    if (m.face_param[t] < -1 || m.face_param[t] >= s->n_params)
                        return fail(ctx, DRT_ERR_INVALID, "scene: face parameter index out of range");
        } else if (sh.type != DRT_SHAPE_PLANE && sh.type != DRT_SHAPE_SPHERE)
            return fail(ctx, DRT_ERR_INVALID, "scene: unknown shape type");
        if (sh.material < -1 || sh.material >= s->n_materials || sh.emitter < -1 || sh.emitter >= s->n_emitters)
            return fail(ctx, DRT_ERR_INVALID, "scene: shape material/emitter index out of range");
    }
    bool any_user_bxdf = false;
    const int n_bxdf_kinds = s->n_kinds != 0 ? s->n_bxdf_kinds : 0;      // (ABI <= 7 callers: the fields behind `meshes` do not exist)
    for (int i = 0; i < s->n_materials; ++i) {
        if (s->materials[i].type == DRT_BXDF_MIRROR)
            continue;                  // no colour parameter
        if (s->materials[i].type >= DRT_BXDF_USER) {
            const int k = s->materials[i].type - DRT_BXDF_USER;
            if (n_bxdf_kinds < 1 || n_bxdf_kinds > DRT_MAX_USER_BXDF_KINDS || !s->bxdf_kinds || k >= n_bxdf_kinds || !s->bxdf_kinds[k].sample_src)
                return fail(ctx, DRT_ERR_INVALID, "scene: unknown material type (a DRT_BXDF_USER + k material needs entry k of drt_scene_desc.bxdf_kinds)");
            any_user_bxdf = true;
        } else
        if (s->materials[i].type != DRT_BXDF_DIFFUSE && s->materials[i].type != DRT_BXDF_SPECULAR)
            return fail(ctx, DRT_ERR_INVALID, "scene: unknown material type");
        if (s->materials[i].param < 0 || s->materials[i].param >= s->n_params)
            return fail(ctx, DRT_ERR_INVALID, "scene: material parameter index out of range");
    }
    for (int i = 0; i < s->n_emitters; ++i)
        if (s->emitters[i].param < 0 || s->emitters[i].param >= s->n_params)
            return fail(ctx, DRT_ERR_INVALID, "scene: emitter parameter index out of range");
    if ((any_user || any_user_bxdf) && any_mesh)
        return fail(ctx, DRT_ERR_UNSUPPORTED, "scene: caller-defined shape or BxDF kinds and a triangle mesh in one scene (the kinds live in the one-launch "
                                              "path kernel compiled for the scene; mesh scenes walk their BVH in kernels of the library's own)");
    // the caller-defined kinds as the header hiprtc compiles them from (drt_prog.h: DRT_USER_SHAPES)
    std::string user_header;
    if (any_user || any_user_bxdf) {
        for (int k = 0; k < DRT_MAX_USER_BXDF_KINDS; ++k) {
            const bool have = k < n_bxdf_kinds;
            char head[320];
            snprintf(head, sizeof head, "// BxDF kind %d: %.64s\ntemplate <typename R> __device__ inline void drt_user_bxdf_%d(const R* p, V3<R> n, V3<R> d, R u1, R u2, "
                                        "V3<R>& wo, R& pdf, R& bs)\n{\n", k, have && s->bxdf_kinds[k].name ? s->bxdf_kinds[k].name : "(none)", k);
            user_header += head;
            user_header += have ? s->bxdf_kinds[k].sample_src : "(void)p; (void)d; (void)u1; (void)u2; wo = n; pdf = R(1); bs = R(0);";
            user_header += "\n}\n";
        }
        for (int k = 0; k < DRT_MAX_USER_KINDS; ++k) {
            const bool have = any_user && k < s->n_kinds;
            char head[320];
            snprintf(head, sizeof head, "// kind %d: %.64s\ntemplate <typename R> __device__ inline bool drt_user_intersect_%d(const R* p, V3<R> o, V3<R> d, R& t)\n{\n",
                     k, have && s->kinds[k].name ? s->kinds[k].name : "(none)", k);
            user_header += head;
            user_header += have ? s->kinds[k].intersect_src : "(void)p; (void)o; (void)d; (void)t; return false;";
            snprintf(head, sizeof head, "\n}\ntemplate <typename R> __device__ inline V3<R> drt_user_normal_%d(const R* p, V3<R> P)\n{\n", k);
            user_header += head;
            user_header += have ? s->kinds[k].normal_src : "(void)p; return P;";
            user_header += "\n}\n";
        }
    }

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
    ctx->n_grad_slots = hf->n_grad_slots;
    ctx->prog_ok = hf->prog_ok != 0;
    ctx->prog_sorted = hf->prog_sorted != 0;
    // The kernels hiprtc made for the PREVIOUS scene's shape kinds are of no use to a scene with other kinds: unload them (a
    // long-lived context that sees scene after scene -- an editor, a fuzzer -- otherwise keeps one loaded module per signature
    // and variant until it is destroyed).  Nothing of this context is running: the streams were waited for above.
    if (memcmp(ctx->prog_sig, sig, sizeof ctx->prog_sig) != 0 || ctx->n_shapes != s->n_shapes || ctx->user_header != user_header) {
        for (int i = 0; i < 2; ++i)
            if (ctx->path_stream[i])
                (void)hipStreamSynchronize(ctx->path_stream[i]);
        for (hipModule_t m : ctx->jit_modules)
            (void)hipModuleUnload(m);
        ctx->jit_modules.clear();
        ctx->jit_fn.clear();
    }
    for (int i = 0; i < 4; ++i)
        ctx->prog_sig[i] = sig[i];
    ctx->user_header = user_header;
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
        int max_face_param = -1;
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
                // the face's colour parameter: its own (face_param) or its material's; a mirror's internal constant either way
                uint32_t cpar = DRT_ID_NONE;
                if (mat >= 0) {
                    const bool mirror = s->materials[mat].type == DRT_BXDF_MIRROR;
                    cpar = mirror ? (uint32_t)s->n_params
                                  : (uint32_t)((m.face_param && m.face_param[k] >= 0) ? m.face_param[k] : s->materials[mat].param);
                    if (!mirror)
                        max_face_param = std::max(max_face_param, (int)cpar);
                }
                t.ids = cpar | ((uint32_t)(mat < 0 ? 0xFF : mat) << 16) | ((uint32_t)(sh.emitter < 0 ? 0xFF : sh.emitter) << 24);
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
            if (built.stack_need > DRT_BVH_STACK || built.nodes.size() >= ((size_t)1 << 24))      // not even a balanced tree fits (> ~2 M triangles)
                return fail(ctx, DRT_ERR_UNSUPPORTED, "scene: the BVH of this mesh needs a deeper traversal stack than the device kernel has");
            if ((rc = upload_bvh<float>(ctx, built, tris, &ctx->bvh_f)) != DRT_OK) return rc;
            if ((rc = upload_bvh<double>(ctx, built, tris, &ctx->bvh_d)) != DRT_OK) return rc;
            ctx->has_mesh = true;
            ctx->max_colour_param = std::max(ctx->max_colour_param, max_face_param);
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
    ctx->emissive_bxdf = false;
    for (int i = 0; i < s->n_shapes; ++i)
        if (s->shapes[i].type != DRT_SHAPE_MESH && s->shapes[i].material >= 0 && s->shapes[i].emitter >= 0)
            ctx->emissive_bxdf = true;
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

// new parameter values -> both compute types' vectors, in stream order (one launch that reads the context's pinned copy;
// grid-stride, coalesced: with an albedo per face -- 152,652 doubles -- a single block spent a millisecond on round trips over
// the link)
__global__ void __launch_bounds__(DRT_BLOCK) k_set_params(const double* __restrict__ h_params, int n, float* __restrict__ pf, double* __restrict__ pd)
{
    for (int i = blockIdx.x * DRT_BLOCK + threadIdx.x; i < n; i += gridDim.x * DRT_BLOCK) {
        const double v = h_params[i];
        pf[i] = (float)v;
        pd[i] = v;
    }
}

int update_params_one(drt_hip_ctx* ctx, const double* params)
{
    if (!ctx)
        return DRT_ERR_INVALID;
    if (!ctx->has_scene)
        return fail(ctx, DRT_ERR_NO_SCENE, "update_params before upload_scene");
    if (!params)
        return fail(ctx, DRT_ERR_INVALID, "params is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->n_user_params * 3;   // internal constants keep their values
    if (n == 0)
        return DRT_OK;
    // (everything enqueued so far has read the old values by the time the pinned copy is overwritten: renders that do not
    //  wait -- device pointers, asynchronous frames -- are the only ones that can still be running here)
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2; ++i)
        if (ctx->path_stream[i])
            HIPCHK(ctx, hipStreamSynchronize(ctx->path_stream[i]));
    if (ctx->h_params_cap < n) {
        if (ctx->h_params)
            (void)hipHostFree(ctx->h_params);
        ctx->h_params = nullptr;
        ctx->h_params_cap = 0;
        HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_params, n * sizeof(double)));
        ctx->h_params_cap = n;
    }
    memcpy(ctx->h_params, params, n * sizeof(double));
    const unsigned blocks = (unsigned)std::min<size_t>((n + DRT_BLOCK - 1) / DRT_BLOCK, (size_t)ctx->n_cu * 2);
    hipLaunchKernelGGL(k_set_params, dim3(blocks ? blocks : 1), dim3(DRT_BLOCK), 0, ctx->stream, (const double*)ctx->h_params, (int)n, ctx->d_params_f, ctx->d_params_d);
    HIPCHK(ctx, hipGetLastError());
    // The install runs in the CONTEXT's stream and the call does not wait for it.  Frames that overlap launch their path kernels
    // on streams of their own (path_stream[lane]); those must not start before the new values are in place: the next frame of
    // either lane waits for this event first (path_batch).
    if (ctx->ev_params) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_params, ctx->stream));
        ctx->params_pending[0] = ctx->params_pending[1] = true;
    } else
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // (a context without the overlap machinery: the old, blocking form)
    return DRT_OK;
}


} // namespace
