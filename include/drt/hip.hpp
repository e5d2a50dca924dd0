// drt/hip.hpp -- host glue above the C ABI (include/drt_hip.h): the batched, device-side
// replacement of the reference's pixel x sample loop (src/render.cpp:72-86).
//
//   drt::hip::render(scene, cam, tracer, spp, img [, options [, adjoint]])
//
// walks a drt::Scene<T> through the additive describe()/kind()/parameter() hooks, deduplicates the
// scene parameters by tape-node identity (handles share nodes: `white` feeds two materials in
// render.cpp:28,34-35), uploads the POD scene, renders on one or several MI355X devices and, when
// options.backward is set, ADDS the returned gradients into param.grad() -- the accumulate
// semantics of VariableNode::backward (vector.hpp:185-188).  Several devices = ONE group context
// (drt_hip_create_group): the library deals the row bands to the devices, runs them side by side and
// sums the gradient vector across them with a single RCCL all-reduce; this header adds nothing.
//
// No CPU fallback: if libdrt_hip.so cannot create a context this throws std::runtime_error.
#pragma once

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../drt_hip.h"
#include "camera.hpp"
#include "mesh.hpp"
#include "pathtracer.hpp"

namespace drt { namespace hip {

struct Options {
    bool backward = false;          // also back-propagate (render.cpp:80, commented out there)
    bool unbiased = false;          // backward with the unbiased integration operator (integrate.hpp:39-52)
    bool sample_loss_l2 = false;    // `adjoint` is a TARGET image: every sample is back-propagated through a loss of its own,
                                    // |radiance - target|^2 (README.md:93-98 with loss_func = squared error; DRT_RENDER_LOSS_L2)
    uint32_t seed = 1;
    int max_depth = 0;              // 0 = library default (64)
    std::vector<int> devices = {0}; // pixel-row bands are dealt round-robin to these devices (several: one group
                                    // context, gradients reduced across them by RCCL inside the library)
    int band_rows = 16;
    bool f64 = false;               // verification mode: compute in double on the device
    long long batch_paths = 0;
    int bounces_per_launch = 0;     // 0 = automatic; 1 = one shade launch per bounce (drt_hip.h)
    bool reuse_context = true;      // keep the device context (and its gigabytes of queues) between calls
};

struct Stats {
    unsigned long long paths = 0, segments = 0;
    unsigned long long capped_paths = 0;   // paths Options::max_depth cut short (the reference has no cap)
    double ms = 0;
};

template <typename T>
struct FlatScene {
    std::vector<drt_shape_desc> shapes;
    std::vector<drt_material_desc> materials;
    std::vector<drt_emitter_desc> emitters;
    std::vector<double> params;
    std::vector<uint8_t> requires_grad;
    std::vector<Vector<T, 3, true>> handles;   // one per parameter, sharing the user's nodes
    std::vector<drt_mesh_desc> meshes;
    std::vector<std::vector<double>> mesh_vertices;
    std::vector<std::vector<uint32_t>> mesh_indices;
    std::vector<drt_shape_kind_desc> kinds;    // caller-defined shape kinds (ShapeKind::User), by kind_name
    std::vector<double> user_params;           // n_shapes x 4: values 4..7 of the shapes' records
    std::vector<drt_bxdf_kind_desc> bxdf_kinds;   // caller-defined BxDF kinds (BxDFKind::User), by name
    std::vector<double> user_bxdf_params;      // n_materials: value 1 of the materials' records

    drt_scene_desc desc() const
    {
        drt_scene_desc d;
        d.n_shapes = (int32_t)shapes.size();
        d.n_materials = (int32_t)materials.size();
        d.n_emitters = (int32_t)emitters.size();
        d.n_params = (int32_t)requires_grad.size();
        d.shapes = shapes.data();
        d.materials = materials.data();
        d.emitters = emitters.data();
        d.params = params.data();
        d.requires_grad = requires_grad.data();
        d.n_meshes = (int32_t)meshes.size();
        d.n_kinds = kinds.empty() ? (bxdf_kinds.empty() ? 0 : -1) : (int32_t)kinds.size();
        d.meshes = meshes.data();
        d.kinds = kinds.data();
        d.user_params = kinds.empty() ? nullptr : user_params.data();
        d.n_bxdf_kinds = (int32_t)bxdf_kinds.size();
        d.reserved2 = 0;
        d.bxdf_kinds = bxdf_kinds.data();
        d.user_bxdf_params = bxdf_kinds.empty() ? nullptr : user_bxdf_params.data();
        return d;
    }

    // Everything but the parameter VALUES (FNV-1a over the records and the mesh data): two scenes with the
    // same key differ at most in their parameters, which drt_hip_update_params replaces without
    // re-uploading the geometry or rebuilding the BVH.
    uint64_t topology_key() const
    {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, std::size_t n) {
            const unsigned char* b = static_cast<const unsigned char*>(p);
            for (std::size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
        };
        auto mix_n = [&](uint64_t v) { mix(&v, sizeof v); };
        mix_n(shapes.size());
        for (const drt_shape_desc& sh : shapes) {       // field by field: the records have padding
            mix(&sh.type, sizeof sh.type); mix(&sh.material, sizeof sh.material); mix(&sh.emitter, sizeof sh.emitter);
            mix(&sh.mesh, sizeof sh.mesh); mix(sh.p, sizeof sh.p);
        }
        mix_n(kinds.size());
        for (const drt_shape_kind_desc& k : kinds) {
            mix(k.intersect_src, std::strlen(k.intersect_src));
            mix(k.normal_src, std::strlen(k.normal_src));
        }
        if (!kinds.empty())
            mix(user_params.data(), user_params.size() * sizeof(double));
        mix_n(bxdf_kinds.size());
        for (const drt_bxdf_kind_desc& k : bxdf_kinds)
            mix(k.sample_src, std::strlen(k.sample_src));
        if (!bxdf_kinds.empty())
            mix(user_bxdf_params.data(), user_bxdf_params.size() * sizeof(double));
        mix_n(materials.size());
        for (const drt_material_desc& m : materials) {
            mix(&m.type, sizeof m.type); mix(&m.param, sizeof m.param); mix(&m.exponent, sizeof m.exponent);
        }
        mix_n(emitters.size());
        for (const drt_emitter_desc& e : emitters)
            mix(&e.param, sizeof e.param);
        mix_n(requires_grad.size());
        mix(requires_grad.data(), requires_grad.size());
        mix_n(meshes.size());
        for (std::size_t i = 0; i < meshes.size(); ++i) {
            mix(mesh_vertices[i].data(), mesh_vertices[i].size() * sizeof(double));
            mix(mesh_indices[i].data(), mesh_indices[i].size() * sizeof(uint32_t));
            mix_n(meshes[i].face_material ? 1 : 0);
            if (meshes[i].face_material)
                mix(meshes[i].face_material, (std::size_t)meshes[i].n_triangles * sizeof(int32_t));
        }
        return h;
    }
};

template <typename T>
inline FlatScene<T> flatten(const Scene<T>& scene)
{
    FlatScene<T> f;
    std::map<const void*, int> param_of, material_of, emitter_of;
    auto param_index = [&](const Vector<T, 3, true>& h) {
        auto it = param_of.find(h.id());
        if (it != param_of.end())
            return it->second;
        const int idx = (int)f.handles.size();
        param_of[h.id()] = idx;
        f.handles.push_back(h);
        for (int c = 0; c < 3; ++c)
            f.params.push_back(double(real(h[c])));
        f.requires_grad.push_back(h.requires_grad() ? 1 : 0);
        return idx;
    };
    for (Shape<T>* shape : scene) {
        const ShapeRecord rec = shape->describe();
        drt_shape_desc sd{};
        if (rec.kind == ShapeKind::Plane)
            sd.type = DRT_SHAPE_PLANE;
        else if (rec.kind == ShapeKind::Sphere)
            sd.type = DRT_SHAPE_SPHERE;
        else if (rec.kind == ShapeKind::Mesh) {
            auto* mesh = dynamic_cast<Mesh<T>*>(shape);
            if (!mesh)
                throw std::runtime_error("drt::hip: ShapeKind::Mesh reported by a shape that is not drt::Mesh");
            sd.type = DRT_SHAPE_MESH;
            sd.mesh = (int32_t)f.mesh_vertices.size();
            std::vector<double> vs;
            for (const auto& v : mesh->vertices())
                for (int c = 0; c < 3; ++c)
                    vs.push_back(double(real(v[c])));
            std::vector<uint32_t> is;
            for (const auto& t : mesh->triangles())
                for (int c = 0; c < 3; ++c)
                    is.push_back(t[c]);
            f.mesh_vertices.push_back(std::move(vs));
            f.mesh_indices.push_back(std::move(is));
        } else if (rec.kind == ShapeKind::User) {
            // any other analytic shape: its own intersect / normal, compiled into the scene's path kernel (ABI v8)
            if (!rec.kind_name || !rec.intersect_src || !rec.normal_src)
                throw std::runtime_error("drt::hip: a ShapeKind::User record needs kind_name, intersect_src and normal_src");
            int k = -1;
            for (std::size_t i = 0; i < f.kinds.size(); ++i)
                if (std::strcmp(f.kinds[i].name, rec.kind_name) == 0)
                    k = (int)i;
            if (k < 0) {
                if (f.kinds.size() >= DRT_MAX_USER_KINDS)
                    throw std::runtime_error("drt::hip: more caller-defined shape kinds in one scene than the device path takes (DRT_MAX_USER_KINDS)");
                drt_shape_kind_desc kd{rec.kind_name, rec.intersect_src, rec.normal_src};
                k = (int)f.kinds.size();
                f.kinds.push_back(kd);
            }
            sd.type = DRT_SHAPE_USER;
            sd.mesh = k;
        } else
            throw std::runtime_error("drt::hip: shape type has no device record (describe() reports neither one of the library's kinds "
                                     "nor ShapeKind::User with its source)");
        for (int i = 0; i < 4; ++i)
            sd.p[i] = rec.p[i];
        for (int i = 0; i < 4; ++i)
            f.user_params.push_back(rec.q[i]);
        sd.material = -1;
        sd.emitter = -1;
        if (BxDF<T>* b = shape->bxdf()) {
            auto it = material_of.find(b);
            if (it == material_of.end()) {
                drt_material_desc md{};
                if (b->kind() == BxDFKind::Diffuse)
                    md.type = DRT_BXDF_DIFFUSE;
                else if (b->kind() == BxDFKind::Specular)
                    md.type = DRT_BXDF_SPECULAR;
                else if (b->kind() == BxDFKind::Mirror)
                    md.type = DRT_BXDF_MIRROR;
                else if (b->kind() == BxDFKind::User) {
                    // any other BxDF of the form colour x scalar: its own sample-and-evaluate body, compiled into the scene's path kernel
                    if (!b->device_kind_name() || !b->device_sample_src() || !b->parameter())
                        throw std::runtime_error("drt::hip: a BxDFKind::User material needs device_kind_name(), device_sample_src() and parameter()");
                    int k = -1;
                    for (std::size_t i = 0; i < f.bxdf_kinds.size(); ++i)
                        if (std::strcmp(f.bxdf_kinds[i].name, b->device_kind_name()) == 0)
                            k = (int)i;
                    if (k < 0) {
                        if (f.bxdf_kinds.size() >= DRT_MAX_USER_BXDF_KINDS)
                            throw std::runtime_error("drt::hip: more caller-defined BxDF kinds in one scene than the device path takes (DRT_MAX_USER_BXDF_KINDS)");
                        drt_bxdf_kind_desc kd{b->device_kind_name(), b->device_sample_src()};
                        k = (int)f.bxdf_kinds.size();
                        f.bxdf_kinds.push_back(kd);
                    }
                    md.type = DRT_BXDF_USER + k;
                } else
                    throw std::runtime_error("drt::hip: BxDF type has no device record (kind() reports neither one of the library's kinds "
                                             "nor BxDFKind::User with its source)");
                md.param = md.type == DRT_BXDF_MIRROR ? -1 : param_index(*b->parameter());
                md.exponent = b->exponent();
                it = material_of.emplace(b, (int)f.materials.size()).first;
                f.materials.push_back(md);
                f.user_bxdf_params.push_back(b->value1());
            }
            sd.material = it->second;
        }
        if (Emitter<T>* e = shape->emitter()) {
            auto it = emitter_of.find(e);
            if (it == emitter_of.end()) {
                auto* area = dynamic_cast<AreaEmitter<T>*>(e);
                if (!area)
                    throw std::runtime_error("drt::hip: emitter type has no device record");
                drt_emitter_desc ed{};
                ed.param = param_index(area->parameter());
                it = emitter_of.emplace(e, (int)f.emitters.size()).first;
                f.emitters.push_back(ed);
            }
            sd.emitter = it->second;
        }
        f.shapes.push_back(sd);
    }
    for (std::size_t m = 0; m < f.mesh_vertices.size(); ++m) {   // pointers taken once the vectors stopped growing
        drt_mesh_desc md{};
        md.n_vertices = (int32_t)(f.mesh_vertices[m].size() / 3);
        md.n_triangles = (int32_t)(f.mesh_indices[m].size() / 3);
        md.vertices = f.mesh_vertices[m].data();
        md.indices = f.mesh_indices[m].data();
        md.face_material = nullptr;
        md.face_param = nullptr;
        f.meshes.push_back(md);
    }
    return f;
}

template <typename T>
inline drt_camera_desc describe(const Camera<T>& cam)
{
    drt_camera_desc c{};
    c.width = (int32_t)cam.width();
    c.height = (int32_t)cam.height();
    c.vfov = cam.vfov();
    for (int i = 0; i < 3; ++i) {
        c.eye[i] = double(real(cam.eye()[i]));
        c.forward[i] = double(real(cam.forward()[i]));
        c.right[i] = double(real(cam.right()[i]));
        c.up[i] = double(real(cam.up()[i]));
    }
    return c;
}

class Context {
public:
    explicit Context(int device) : Context(std::vector<int>{device}) {}
    // one device: a plain context; several: a group context (the library owns the RCCL communicators)
    explicit Context(const std::vector<int>& devices)
    {
        if (devices.empty())
            throw std::runtime_error("drt::hip::Context: no device given");
        const int rc = devices.size() == 1 ? drt_hip_create(devices[0], &m_ctx)
                                           : drt_hip_create_group(devices.data(), (int)devices.size(), &m_ctx);
        if (rc != DRT_OK) {
            std::string list;
            for (int d : devices)
                list += (list.empty() ? "" : ", ") + std::to_string(d);
            throw std::runtime_error("drt_hip_create" + std::string(devices.size() == 1 ? "" : "_group") + "(device " + list +
                                     ") failed with status " + std::to_string(rc) + " (no HIP device? there is no CPU fallback)");
        }
    }
    ~Context() { drt_hip_destroy(m_ctx); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    drt_hip_ctx* get() const { return m_ctx; }
    void check(int rc, const char* what) const
    {
        if (rc != DRT_OK)
            throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + drt_hip_last_error(m_ctx));
    }
    std::mutex& mutex() { return m_mutex; }

    // upload the scene, or -- when only parameter values changed since this context's last upload -- just them
    template <typename T>
    void set_scene(const FlatScene<T>& flat)
    {
        const uint64_t key = flat.topology_key();
        if (m_has_scene && key == m_scene_key) {
            if (flat.params == m_params)       // nothing changed (drt_hip_update_params waits for the frames in flight)
                return;
            check(drt_hip_update_params(m_ctx, flat.params.data()), "drt_hip_update_params");
            m_params = flat.params;
            return;
        }
        const drt_scene_desc sd = flat.desc();
        m_has_scene = false;
        check(drt_hip_upload_scene(m_ctx, &sd), "drt_hip_upload_scene");
        m_scene_key = key;
        m_params = flat.params;
        m_has_scene = true;
    }

private:
    drt_hip_ctx* m_ctx = nullptr;
    uint64_t m_scene_key = 0;
    std::vector<double> m_params;      // the parameter values the device holds
public:
    // host buffers of the (at most DRT_HIP_FRAMES_IN_FLIGHT) frames in flight: kept, so that a frame costs no 3 MB allocation
    std::vector<float> frame_pool[DRT_HIP_FRAMES_IN_FLIGHT];
    std::vector<double> grad_pool[DRT_HIP_FRAMES_IN_FLIGHT];
    bool slot_in_flight[DRT_HIP_FRAMES_IN_FLIGHT] = {};   // the slot's buffers belong to a frame that has not been collected
    unsigned submitted = 0;
    // The frame buffer of the SYNCHRONOUS call, kept between calls and -- on a plain context -- pinned (drt_hip_pin_host): the
    // device's finishing kernel then writes a frame straight into it; no 3 MB allocation, staging copy or memcpy per call.
    float* sync_frame(std::size_t n_floats, bool plain)
    {
        if (m_sync_frame.size() != n_floats) {
            if (m_sync_pinned) {
                (void)drt_hip_unpin_host(m_ctx, m_sync_frame.data());
                m_sync_pinned = false;
            }
            m_sync_frame.assign(n_floats, 0.f);
            if (plain && n_floats)
                m_sync_pinned = drt_hip_pin_host(m_ctx, m_sync_frame.data(), n_floats * sizeof(float)) == DRT_OK;
        }
        return m_sync_frame.data();
    }
private:
    std::vector<float> m_sync_frame;
    bool m_sync_pinned = false;
public:
private:
    bool m_has_scene = false;
    std::mutex m_mutex;        // a context is not thread-safe: pooled ones are locked for the duration of a call
};

// Contexts are kept between calls: creating one (HIP module load, stream) and growing its queues
// (gigabytes of hipMalloc for a 512 x 512 x 64 frame) costs ~0.3 s, a render 2 ms -- an optimisation
// loop around drt::hip::render must not pay that per iteration.  One context per device list (a
// group context for several devices).  They live until release_contexts() or process exit
// (deliberately not destroyed by static destructors: the HIP runtime may be gone by then).
namespace detail {
struct ContextPool {
    std::mutex m;
    std::map<std::vector<int>, Context*> contexts;
};
inline ContextPool& pool()
{
    static ContextPool* p = new ContextPool();
    return *p;
}
} // namespace detail

inline Context& pooled_context(const std::vector<int>& devices)
{
    detail::ContextPool& p = detail::pool();
    std::lock_guard<std::mutex> lock(p.m);
    Context*& c = p.contexts[devices];
    if (!c)
        c = new Context(devices);
    return *c;
}
inline Context& pooled_context(int device) { return pooled_context(std::vector<int>{device}); }

inline void release_contexts()
{
    detail::ContextPool& p = detail::pool();
    std::lock_guard<std::mutex> lock(p.m);
    for (auto& kv : p.contexts)
        delete kv.second;
    p.contexts.clear();
}

// img: width*height row-major (render.cpp:66,82); adjoint: optional per-pixel seed, same layout.
template <typename T>
inline Stats render(const Scene<T>& scene, const Camera<T>& cam, const Pathtracer<T>& tracer, std::size_t spp,
                    Vector<T, 3>* img, const Options& opt = Options(), const Vector<T, 3>* adjoint = nullptr)
{
    FlatScene<T> flat = flatten(scene);
    const drt_camera_desc cd = describe(cam);
    const std::size_t npix = cam.width() * cam.height();
    const int n_dev = (int)opt.devices.size();
    if (n_dev < 1)
        throw std::runtime_error("drt::hip::render: no device given");

    std::vector<float> adj;
    if (adjoint) {
        adj.resize(npix * 3);
        for (std::size_t i = 0; i < npix; ++i)
            for (int c = 0; c < 3; ++c)
                adj[i * 3 + c] = float(real(adjoint[i][c]));
    }
    const std::size_t P = flat.requires_grad.size();
    std::vector<float> own_frame;
    const float* frame = nullptr;
    std::vector<double> grads(P * 3, 0.0);
    drt_hip_stats st{};
    {
        std::unique_ptr<Context> own;
        if (!opt.reuse_context)
            own.reset(new Context(opt.devices));
        Context& ctx = own ? *own : pooled_context(opt.devices);
        std::lock_guard<std::mutex> lock(ctx.mutex());
        ctx.set_scene(flat);
        float* out = nullptr;
        if (own) {
            own_frame.assign(npix * 3, 0.f);
            out = own_frame.data();
        } else
            out = ctx.sync_frame(npix * 3, n_dev == 1);      // (the pooled context's own buffer, pinned on a plain context)
        frame = out;
        drt_render_params rp{};
        rp.spp = (int32_t)spp;
        rp.min_bounces = (int32_t)tracer.min_bounces();
        rp.absorb = tracer.absorb();
        rp.max_depth = opt.max_depth;
        rp.seed = opt.seed;
        rp.shard = 0;
        rp.n_shards = 1;                // a group context deals the bands to its devices itself
        rp.band_rows = opt.band_rows;
        rp.flags = (opt.backward ? DRT_RENDER_BACKWARD : 0u) | (opt.f64 ? DRT_RENDER_F64 : 0u) |
                   (opt.backward && opt.unbiased ? DRT_RENDER_UNBIASED : 0u) |
                   (opt.backward && opt.sample_loss_l2 ? DRT_RENDER_LOSS_L2 : 0u);
        rp.batch_paths = opt.batch_paths;
        rp.bounces_per_launch = opt.bounces_per_launch;
        // n_dev > 1: out_param_grad comes back ALREADY summed over the devices (one ncclAllReduce in the library)
        ctx.check(drt_hip_render(ctx.get(), &cd, &rp, adjoint ? adj.data() : nullptr, out,
                                 opt.backward ? grads.data() : nullptr, &st),
                  "drt_hip_render");
        for (std::size_t i = 0; i < npix; ++i)                 // (under the context's lock: the buffer is the context's)
            for (int c = 0; c < 3; ++c)
                img[i][c] = T(frame[i * 3 + c]);
    }
    Stats total;
    total.paths = st.paths;
    total.segments = st.segments;
    total.capped_paths = st.capped_paths;
    total.ms = st.ms_total;
    if (opt.backward) {
        for (std::size_t p = 0; p < P; ++p) {
            if (!flat.requires_grad[p])
                continue;
            Vector<T, 3> g(T(0));
            for (int c = 0; c < 3; ++c)
                g[c] = T(grads[p * 3 + c]);
            flat.handles[p].grad() += g;          // accumulate, like m_grad += grad
        }
    }
    return total;
}

// ---- frames in flight ----------------------------------------------------------------------------------
// render() returns with the frame in `img`: a device-to-host copy and a wait per call, the GPU idle meanwhile.  A loop
// that renders frame after frame (several views or mini-batches per optimisation step, a turntable) submits the next
// frame before it collects the previous one:
//     auto a = drt::hip::submit(scene, cam, tracer, spp, img_a, opt);
//     auto b = drt::hip::submit(scene, cam, tracer, spp, img_b, opt);   // at most four in flight per device context
//     a.get();  b.get();             // img_* filled, gradients ADDED into param.grad() (vector.hpp:185-188) at get()
// (drt_hip_render_async / drt_hip_wait: frame i's results travel to a pinned block while frame i + 1's kernels run.)
// One device (opt.devices[0]); the frames of one context belong to one thread; the scene's geometry must not change
// between submit and get (parameter values may: a submit uploads them).
template <typename T>
class Pending {
public:
    Pending() = default;
    Pending(Pending&& o) noexcept { *this = std::move(o); }
    // A frame that is dropped without get() -- an exception between submit and get, a handle that is overwritten -- is still
    // waited for and its results discarded: its slot of the (process-wide, pooled) context would otherwise stay in flight for
    // ever and every later render on that device would be refused.
    ~Pending() { discard(); }
    Pending& operator=(Pending&& o) noexcept
    {
        if (this == &o)
            return *this;
        discard();
        m_ctx = o.m_ctx; o.m_ctx = nullptr;            // (the source no longer owns a frame)
        m_slot = o.m_slot;
        m_ticket = o.m_ticket; m_img = o.m_img; m_backward = o.m_backward;
        m_frame = o.m_frame; m_grads = o.m_grads; m_npix = o.m_npix;
        m_requires_grad = std::move(o.m_requires_grad); m_handles = std::move(o.m_handles);
        return *this;
    }
    Pending(const Pending&) = delete;
    Pending& operator=(const Pending&) = delete;
    bool valid() const { return m_ctx != nullptr; }
    // wait for the frame, hand it over, accumulate its gradients
    Stats get()
    {
        if (!m_ctx)
            throw std::runtime_error("drt::hip::Pending::get: no frame");
        drt_hip_stats st{};
        Context* ctx = m_ctx;
        m_ctx = nullptr;
        std::lock_guard<std::mutex> lock(ctx->mutex());
        const int rc = drt_hip_wait(ctx->get(), m_ticket, &st);
        ctx->slot_in_flight[m_slot] = false;
        ctx->check(rc, "drt_hip_wait");
        const std::size_t npix = m_npix;
        for (std::size_t i = 0; i < npix; ++i)
            for (int c = 0; c < 3; ++c)
                m_img[i][c] = T(m_frame[i * 3 + c]);
        if (m_backward)
            for (std::size_t p = 0; p < m_handles.size(); ++p) {
                if (!m_requires_grad[p])
                    continue;
                Vector<T, 3> g(T(0));
                for (int c = 0; c < 3; ++c)
                    g[c] = T(m_grads[p * 3 + c]);
                m_handles[p].grad() += g;
            }
        Stats total;
        total.paths = st.paths;
        total.segments = st.segments;
        total.capped_paths = st.capped_paths;
        total.ms = st.ms_total;
        return total;
    }

private:
    void discard() noexcept
    {
        if (!m_ctx)
            return;
        Context* ctx = m_ctx;
        m_ctx = nullptr;
        std::lock_guard<std::mutex> lock(ctx->mutex());
        (void)drt_hip_wait(ctx->get(), m_ticket, nullptr);      // (into the pool's buffers; nothing is accumulated)
        ctx->slot_in_flight[m_slot] = false;
    }
    template <typename U>
    friend Pending<U> submit(const Scene<U>&, const Camera<U>&, const Pathtracer<U>&, std::size_t, Vector<U, 3>*, const Options&,
                             const Vector<U, 3>*);
    Context* m_ctx = nullptr;
    unsigned m_slot = 0;
    uint64_t m_ticket = 0;
    Vector<T, 3>* m_img = nullptr;
    bool m_backward = false;
    float* m_frame = nullptr;                   // written by drt_hip_wait: buffers of the context's pool (a set per frame in flight)
    double* m_grads = nullptr;
    std::size_t m_npix = 0;
    std::vector<uint8_t> m_requires_grad;
    std::vector<Vector<T, 3, true>> m_handles;
};

template <typename T>
inline Pending<T> submit(const Scene<T>& scene, const Camera<T>& cam, const Pathtracer<T>& tracer, std::size_t spp,
                         Vector<T, 3>* img, const Options& opt = Options(), const Vector<T, 3>* adjoint = nullptr)
{
    if (opt.devices.size() != 1)
        throw std::runtime_error("drt::hip::submit: one device (frames in flight are per device context)");
    FlatScene<T> flat = flatten(scene);
    const drt_camera_desc cd = describe(cam);
    const std::size_t npix = cam.width() * cam.height();
    std::vector<float> adj;
    if (adjoint) {
        adj.resize(npix * 3);
        for (std::size_t i = 0; i < npix; ++i)
            for (int c = 0; c < 3; ++c)
                adj[i * 3 + c] = float(real(adjoint[i][c]));
    }
    Pending<T> f;
    f.m_img = img;
    f.m_backward = opt.backward;
    f.m_requires_grad = flat.requires_grad;
    f.m_handles = flat.handles;
    Context& ctx = pooled_context(opt.devices);
    std::lock_guard<std::mutex> lock(ctx.mutex());
    const unsigned slot = ctx.submitted % DRT_HIP_FRAMES_IN_FLIGHT;
    // (before the slot's buffers are touched: a frame still in flight is written into them at its get())
    if (ctx.slot_in_flight[slot])
        throw std::runtime_error("drt::hip::submit: four frames are in flight on this device -- get() the oldest one first");
    if (ctx.frame_pool[slot].size() < npix * 3) ctx.frame_pool[slot].resize(npix * 3);
    if (ctx.grad_pool[slot].size() < flat.requires_grad.size() * 3) ctx.grad_pool[slot].resize(flat.requires_grad.size() * 3);
    f.m_frame = ctx.frame_pool[slot].data();
    f.m_grads = ctx.grad_pool[slot].data();
    f.m_npix = npix;
    ctx.set_scene(flat);
    drt_render_params rp{};
    rp.spp = (int32_t)spp;
    rp.min_bounces = (int32_t)tracer.min_bounces();
    rp.absorb = tracer.absorb();
    rp.max_depth = opt.max_depth;
    rp.seed = opt.seed;
    rp.n_shards = 1;
    rp.band_rows = opt.band_rows;
    rp.flags = (opt.backward ? DRT_RENDER_BACKWARD : 0u) | (opt.f64 ? DRT_RENDER_F64 : 0u) |
               (opt.backward && opt.unbiased ? DRT_RENDER_UNBIASED : 0u) |
               (opt.backward && opt.sample_loss_l2 ? DRT_RENDER_LOSS_L2 : 0u);
    rp.batch_paths = opt.batch_paths;
    rp.bounces_per_launch = opt.bounces_per_launch;
    ctx.check(drt_hip_render_async(ctx.get(), &cd, &rp, adjoint ? adj.data() : nullptr, f.m_frame,
                                   opt.backward ? f.m_grads : nullptr, &f.m_ticket),
              "drt_hip_render_async");
    ++ctx.submitted;
    ctx.slot_in_flight[slot] = true;
    f.m_slot = slot;
    f.m_ctx = &ctx;
    return f;
}

// Per-pixel gradient image of ONE parameter (the figure of the reference's README.md:142-145):
// gimg[pixel] = mean over the pixel's samples of d(seed . radiance)/d param.  Single device.
template <typename T>
inline Stats render_gradient_image(const Scene<T>& scene, const Camera<T>& cam, const Pathtracer<T>& tracer,
                                   std::size_t spp, const Vector<T, 3, true>& param, Vector<T, 3>* img,
                                   Vector<T, 3>* gimg, const Options& opt = Options())
{
    FlatScene<T> flat = flatten(scene);
    int index = -1;
    for (std::size_t p = 0; p < flat.handles.size(); ++p)
        if (flat.handles[p].id() == param.id())
            index = (int)p;
    if (index < 0)
        throw std::runtime_error("drt::hip::render_gradient_image: the parameter is not used by the scene");
    const drt_camera_desc cd = describe(cam);
    const std::size_t npix = cam.width() * cam.height();
    std::vector<float> rgb(npix * 3, 0.f), grad(npix * 3, 0.f);
    std::unique_ptr<Context> own;
    if (!opt.reuse_context)
        own.reset(new Context(opt.devices.empty() ? 0 : opt.devices[0]));
    Context& ctx = own ? *own : pooled_context(opt.devices.empty() ? 0 : opt.devices[0]);
    std::lock_guard<std::mutex> lock(ctx.mutex());
    ctx.set_scene(flat);
    drt_render_params rp{};
    rp.spp = (int32_t)spp;
    rp.min_bounces = (int32_t)tracer.min_bounces();
    rp.absorb = tracer.absorb();
    rp.max_depth = opt.max_depth;
    rp.seed = opt.seed;
    rp.flags = opt.f64 ? DRT_RENDER_F64 : 0u;
    rp.batch_paths = opt.batch_paths;
    drt_hip_stats st{};
    ctx.check(drt_hip_render_gradient_image(ctx.get(), &cd, &rp, index, nullptr, rgb.data(), grad.data(), &st),
              "drt_hip_render_gradient_image");
    for (std::size_t i = 0; i < npix; ++i)
        for (int c = 0; c < 3; ++c) {
            if (img) img[i][c] = T(rgb[i * 3 + c]);
            gimg[i][c] = T(grad[i * 3 + c]);
        }
    Stats out;
    out.paths = st.paths;
    out.segments = st.segments;
    out.ms = st.ms_total;
    return out;
}

} } // namespace drt::hip
