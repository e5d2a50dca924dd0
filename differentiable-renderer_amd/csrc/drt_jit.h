// drt_jit.h -- run-time specialisation of k_path (host side; included by drt_hip.hip only).
//
// k_path's closest-hit test is fastest when the KINDS of the scene's shapes (axis plane, general plane, sphere) are template
// constants: the shape loop unrolls, the records sit in scalar registers, the bounce loop has no branch, no load and no wait
// for the scene (drt_prog.h: closest_hit_sig).  The library carries that instantiation for the reference's own scene
// (render.cpp:39-47); for every other analytic scene it is made HERE, with hiprtc, from the very headers the library was
// built from (embedded as strings: embed_sources.py) -- `k_path<float, false, 4, 3, KindSig<the scene's words>, false>` as
// a name expression, ~0.3-0.7 s per variant on the box's host, cached per process by that name.  The result is
// bit-identical to the kind-sorted program it replaces (same record arithmetic: prog_t; tests/test_gpu_jit.py), so WHEN a
// scene gets its own program is a matter of cost only: once it has rendered enough for the compile to pay
// (drt_hip.hip: jit_wanted), at once with DRT_HIP_JIT=force, never with DRT_HIP_JIT=0.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>
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

inline std::mutex& mutex() { static std::mutex m; return m; }
inline std::map<std::string, Code>& cache() { static std::map<std::string, Code> c; return c; }

// compile the instantiation `name_expr` of a kernel template of drt_path.h for `arch` (process-wide cache; thread-safe:
// the members of a group context launch from threads of their own)
inline const Code& compile(const std::string& arch, const std::string& name_expr)
{
    std::lock_guard<std::mutex> lock(mutex());
    const std::string key = arch + "|" + name_expr;
    auto it = cache().find(key);
    if (it != cache().end())
        return it->second;
    Code& c = cache()[key];
    const auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, "#include \"drt_path.h\"\n", "drt_jit.hip", drt_jit_n_headers, drt_jit_header_srcs,
                                         drt_jit_header_names);
    if (r != HIPRTC_SUCCESS) {
        c.log = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return c;
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
        return c;
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
    return c;
}

// "KindSig<0x...ull, 0x...ull, 0x...ull, 0x...ull, n>"
inline std::string sig_type(const unsigned long long sig[4], int n_shapes)
{
    char buf[160];
    snprintf(buf, sizeof buf, "KindSig<0x%llxull, 0x%llxull, 0x%llxull, 0x%llxull, %d>", sig[0], sig[1], sig[2], sig[3], n_shapes);
    return buf;
}

} // namespace drt_jit
