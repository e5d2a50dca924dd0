// drt_jit.h -- run-time specialisation of k_path (host side; included by drt_hip.hip only).
//
// k_path's closest-hit test is fastest when the KINDS of the scene's shapes (axis plane, general plane, sphere) are template
// constants: the shape loop unrolls, the records sit in scalar registers, the bounce loop has no branch, no load and no wait
// for the scene (drt_prog.h: closest_hit_sig).  The library carries that instantiation for the reference's own scene
// (render.cpp:39-47); for every other analytic scene it is made HERE, with hiprtc, from the very headers the library was
// built from (embedded as strings: embed_sources.py) -- `k_path<float, false, 4, 3, KindSig<the scene's words>, false>` as
// a name expression, ~0.3-0.7 s per variant on the box's host, cached per process by that name.  The result is
// bit-identical to the kind-sorted program it replaces (same record arithmetic: prog_t; tests/test_gpu_jit.py), so WHEN a
// scene gets its own program is a matter of cost only: once it has rendered enough for the compile to pay -- then on a
// thread of its own, no frame waits for the compiler (poll) --, at once and waited for with DRT_SPECIALISE_NOW /
// DRT_HIP_JIT=force (compile), never with DRT_HIP_JIT=0.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace drt_jit {

#include "drt_jit_sources.inc"

struct Code {
    bool ok = false;
    std::vector<char> bin;      // the code object
    std::string lowered;        // mangled name of the instantiation
    std::string log;            // compiler output / error
    double ms = 0;              // compile time
};

// the hiprtc work itself (no lock held; ~0.3-0.7 s of one host core)
// (user_header: the scene's caller-defined shape kinds as "drt_user_shapes.h" -- drt_prog.h includes it under DRT_USER_SHAPES --,
//  empty: none)
inline void build(const std::string& arch, const std::string& name_expr, const std::string& user_header, Code& c)
{
    const auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog = nullptr;
    std::vector<const char*> srcs(drt_jit_header_srcs, drt_jit_header_srcs + drt_jit_n_headers);
    std::vector<const char*> names(drt_jit_header_names, drt_jit_header_names + drt_jit_n_headers);
    if (!user_header.empty()) {
        srcs.push_back(user_header.c_str());
        names.push_back("drt_user_shapes.h");
    }
    hiprtcResult r = hiprtcCreateProgram(&prog, user_header.empty() ? "#include \"drt_path.h\"\n" : "#define DRT_USER_SHAPES 1\n#include \"drt_path.h\"\n",
                                         "drt_jit.hip", (int)srcs.size(), srcs.data(), names.data());
    if (r != HIPRTC_SUCCESS) {
        c.log = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return;
    }
    r = hiprtcAddNameExpression(prog, name_expr.c_str());
    const std::string arch_opt = "--offload-arch=" + arch;
    // (the options of the library's own build: Makefile / build_native.  NOT -ffp-contract=fast: the HIP default is
    //  fast-honor-pragmas -- contraction where the front end allows it -- while "fast" lets the back end fuse any multiply-add
    //  it sees, inside the inlined math library too: the glossy sampler then differed from the library's own build in the last bit)
    const char* opts[] = {arch_opt.c_str(), "-O3", "-std=c++17", "-fno-slp-vectorize"};
    if (r == HIPRTC_SUCCESS)
        r = hiprtcCompileProgram(prog, (int)(sizeof opts / sizeof opts[0]), opts);
    size_t log_size = 0;
    if (hiprtcGetProgramLogSize(prog, &log_size) == HIPRTC_SUCCESS && log_size > 1) {
        c.log.resize(log_size);
        (void)hiprtcGetProgramLog(prog, &c.log[0]);
    }
    if (r != HIPRTC_SUCCESS) {
        c.log = std::string("hiprtc (") + name_expr + "): " + hiprtcGetErrorString(r) + "\n" + c.log;
        (void)hiprtcDestroyProgram(&prog);
        return;
    }
    const char* lowered = nullptr;
    size_t size = 0;
    if (hiprtcGetLoweredName(prog, name_expr.c_str(), &lowered) == HIPRTC_SUCCESS && lowered &&
        hiprtcGetCodeSize(prog, &size) == HIPRTC_SUCCESS && size > 0) {
        c.lowered = lowered;
        c.bin.resize(size);
        c.ok = hiprtcGetCode(prog, c.bin.data()) == HIPRTC_SUCCESS;
    }
    if (!c.ok)
        c.log = std::string("hiprtc (") + name_expr + "): no code object";
    (void)hiprtcDestroyProgram(&prog);
    c.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// build() behind a catch-all: nothing may leave a compile as an exception -- it would leave the cache entry "in progress" for
// ever (every later compile() of that key waits on the condition variable) and cross the C ABI.  A failed allocation is a
// failed compile.
inline void build_guarded(const std::string& arch, const std::string& name_expr, const std::string& user_header, Code& c) noexcept
{
    try {
        build(arch, name_expr, user_header, c);
    } catch (...) {
        c.ok = false;
        try {
            c.bin.clear();
            c.log = "hiprtc (" + name_expr + "): the compile ended in an exception (out of memory?)";
        } catch (...) {
        }
    }
}

// The process-wide cache, by architecture and name expression.  An entry exists from the moment somebody asked for it;
// `done` says whether its compile has finished.  Entries are shared: whoever got one keeps it alive, so the cache may drop
// its own reference -- it holds at most DRT_JIT_CACHE_MAX finished code objects (a process that renders thousands of
// different scenes, a fuzzer, must not keep every one of them; ROCm's own on-disk cache makes a recompile cheap).
#ifndef DRT_JIT_CACHE_MAX
#define DRT_JIT_CACHE_MAX 64
#endif
struct Entry {
    Code code;
    bool done = false;
};
typedef std::shared_ptr<Entry> EntryPtr;
struct State {
    std::mutex m;
    std::condition_variable cv;
    std::map<std::string, EntryPtr> cache;
    std::vector<std::string> order;         // keys in the order they were added (the oldest finished ones go first)
    std::vector<std::thread> workers;       // background compiles (poll); joined when the library is unloaded
    ~State()
    {
        for (std::thread& t : workers)
            if (t.joinable())
                t.join();
    }
    // (called with the lock held, before a new key is added)
    void evict()
    {
        size_t i = 0;
        while (cache.size() >= DRT_JIT_CACHE_MAX && i < order.size()) {
            auto it = cache.find(order[i]);
            if (it != cache.end() && it->second->done) {
                cache.erase(it);
                order.erase(order.begin() + (long)i);
            } else
                ++i;
        }
    }
};
inline State& state() { static State s; return s; }

// compile the instantiation `name_expr` of a kernel template of drt_path.h for `arch` and WAIT for it (thread-safe: the
// members of a group context launch from threads of their own; a compile already running in the background is joined)
// (the cache key carries the caller-defined kinds' source: two scenes with the same signature and other shape code are two kernels)
inline std::string cache_key(const std::string& arch, const std::string& name_expr, const std::string& user_header)
{
    std::string key = arch + "|" + name_expr;
    if (!user_header.empty()) {
        unsigned long long h = 1469598103934665603ull;        // FNV-1a
        for (unsigned char ch : user_header) { h ^= ch; h *= 1099511628211ull; }
        char buf[40];
        snprintf(buf, sizeof buf, "|user:%016llx:%zu", h, user_header.size());
        key += buf;
    }
    return key;
}

inline EntryPtr compile(const std::string& arch, const std::string& name_expr, const std::string& user_header = std::string())
{
    State& st = state();
    const std::string key = cache_key(arch, name_expr, user_header);
    std::unique_lock<std::mutex> lock(st.m);
    auto it = st.cache.find(key);
    if (it != st.cache.end()) {
        EntryPtr e = it->second;
        st.cv.wait(lock, [&e] { return e->done; });
        return e;
    }
    st.evict();
    EntryPtr e = std::make_shared<Entry>();
    st.cache[key] = e;
    st.order.push_back(key);
    lock.unlock();
    build_guarded(arch, name_expr, user_header, e->code);
    lock.lock();
    e->done = true;
    st.cv.notify_all();
    return e;
}

// the same WITHOUT waiting: nullptr while the compile runs -- started on a thread of its own by the first call -- and the
// code once it is there.  What DRT_SPECIALISE_AUTO uses: the frames of a render loop never wait for the compiler, they
// run the kind-sorted program (same results, bit for bit) until the specialised one has arrived.
inline EntryPtr poll(const std::string& arch, const std::string& name_expr, const std::string& user_header = std::string())
{
    State& st = state();
    const std::string key = cache_key(arch, name_expr, user_header);
    std::lock_guard<std::mutex> lock(st.m);
    auto it = st.cache.find(key);
    if (it != st.cache.end())
        return it->second->done ? it->second : EntryPtr();
    st.evict();
    EntryPtr e = std::make_shared<Entry>();
    st.cache[key] = e;
    st.order.push_back(key);
    try {
        st.workers.emplace_back([arch, name_expr, user_header, e, &st] {
            build_guarded(arch, name_expr, user_header, e->code);
            {
                std::lock_guard<std::mutex> l(st.m);
                e->done = true;
            }
            st.cv.notify_all();
        });
    } catch (...) {
        // (no thread to be had: the entry must not stay "in progress" for ever -- whoever asks next gets a failed compile and
        //  renders with the kind-sorted program)
        e->code.ok = false;
        e->code.log = std::string("hiprtc (") + name_expr + "): could not start the compile thread";
        e->done = true;
        st.cv.notify_all();
        return e;
    }
    return EntryPtr();
}

// wait for the background compiles that are still running (a context is being destroyed: the process may be about to exit,
// and the compiler's own state should not be torn down under a running compile)
inline void wait_idle()
{
    State& st = state();
    std::vector<std::thread> done;
    {
        std::lock_guard<std::mutex> lock(st.m);
        done.swap(st.workers);
    }
    for (std::thread& t : done)
        if (t.joinable())
            t.join();
}

// "KindSig<0x...ull, 0x...ull, 0x...ull, 0x...ull, n>"
inline std::string sig_type(const unsigned long long sig[4], int n_shapes)
{
    char buf[160];
    snprintf(buf, sizeof buf, "KindSig<0x%llxull, 0x%llxull, 0x%llxull, 0x%llxull, %d>", sig[0], sig[1], sig[2], sig[3], n_shapes);
    return buf;
}

} // namespace drt_jit
