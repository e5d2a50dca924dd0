// hiprtc_probe.cpp -- does hiprtc compile for gfx950 here (no GPU needed), how long does k_path take, and does the code object
// load and run through hipModule* on the box (not part of the product).
//   hipcc -O2 -std=c++17 tools/hiprtc_probe.cpp -o gpurun_out/hiprtc_probe -lhiprtc
//   gpurun_out/hiprtc_probe <file.hip> <kernel name expression or ""> [-I dir ...]
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 2) { printf("usage: %s file.hip [name-expression] [options...]\n", argv[0]); return 2; }
    std::ifstream f(argv[1]);
    std::stringstream ss; ss << f.rdbuf();
    const std::string src = ss.str();
    const std::string name = argc > 2 ? argv[2] : "";
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize"};
    for (int i = 3; i < argc; ++i) opts.push_back(argv[i]);
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "probe.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { printf("create failed\n"); return 1; }
    if (!name.empty()) hiprtcAddNameExpression(prog, name.c_str());
    const auto t0 = std::chrono::steady_clock::now();
    const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    size_t ls = 0; hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { std::string log(ls, 0); hiprtcGetProgramLog(prog, &log[0]); printf("log: %s\n", log.c_str()); }
    printf("compile: %s in %.0f ms\n", hiprtcGetErrorString(r), ms);
    if (r != HIPRTC_SUCCESS) return 1;
    size_t cs = 0; hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
    printf("code object: %zu bytes\n", cs);
    const char* lowered = nullptr;
    if (!name.empty() && hiprtcGetLoweredName(prog, name.c_str(), &lowered) == HIPRTC_SUCCESS) printf("lowered: %s\n", lowered);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { printf("no device: not loaded\n"); return 0; }
    hipModule_t mod; hipError_t e = hipModuleLoadData(&mod, code.data());
    printf("hipModuleLoadData: %s\n", hipGetErrorString(e));
    if (e == hipSuccess && lowered) {
        hipFunction_t fn; e = hipModuleGetFunction(&fn, mod, lowered);
        printf("hipModuleGetFunction: %s\n", hipGetErrorString(e));
    }
    return 0;
}
