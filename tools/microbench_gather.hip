// microbench_gather.hip -- how fast can a CU serve DIVERGENT 16-byte loads (every lane its own cache line)?  The BVH
// walk's node and triangle fetches are exactly that; this measures the roof they run against (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_gather.hip -o gpurun_out/microbench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every lane walks its own pseudo-random sequence of 64-byte records inside a table of `mask + 1` records;
// CHAIN = 1: the next index depends on the loaded data (a dependent chain, like a tree walk); 0: independent loads;
// WORDS = 16-byte words fetched per record (1..4)
template <int CHAIN, int WORDS>
__global__ void __launch_bounds__(256) k_gather(const uint4* __restrict__ table, uint32_t mask, int iters, uint32_t* __restrict__ out)
{
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const uint4* rec = table + (size_t)(idx & mask) * 4;
        uint4 v = rec[0];
        uint32_t s = v.x ^ v.y ^ v.z ^ v.w;
        if (WORDS > 1) { uint4 u = rec[1]; s ^= u.x ^ u.w; }
        if (WORDS > 2) { uint4 u = rec[2]; s ^= u.y ^ u.z; }
        if (WORDS > 3) { uint4 u = rec[3]; s ^= u.x ^ u.y; }
        acc += s;
        idx = idx * 1664525u + 1013904223u + (CHAIN ? s : 0u);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main()
{
    const size_t max_recs = (size_t)1 << 22;             // 256 MB of 64-byte records
    uint4* table; uint32_t* out;
    CK(hipMalloc(&table, max_recs * 64)); CK(hipMalloc(&out, 16));
    std::vector<uint32_t> h(max_recs * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2246822519u) >> 3;
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 256;
    printf("%-10s %-6s %-6s %-7s %10s %14s %16s\n", "table", "chain", "words", "waves/CU", "ms", "Glane-acc/s", "lane-acc/clk/CU");
    for (size_t kb : {16, 1024, 3072, 65536, 262144}) {          // L1-resident, L2-resident (1 MB, 3 MB), Infinity Cache, HBM
        const uint32_t mask = (uint32_t)(kb * 1024 / 64) - 1;    // (3072 KB: 49152 records -> masked to a power of two below)
        uint32_t m = 1; while (m * 2 - 1 <= mask) m *= 2; m -= 1;
        for (int blocks_per_cu : {4, 8}) {
            const int grid = 256 * blocks_per_cu;
#define RUN(CHAIN, WORDS)                                                                                                  \
            {                                                                                                              \
                float t = time_ms([&] { hipLaunchKernelGGL((k_gather<CHAIN, WORDS>), dim3(grid), dim3(256), 0, 0, table, m, iters, out); }, 5); \
                double acc = (double)grid * 256 * iters * WORDS;                                                          \
                printf("%7zu KB %-6d %-6d %-7d %10.3f %14.1f %16.2f\n", (size_t)(m + 1) * 64 / 1024, CHAIN, WORDS, blocks_per_cu * 4, t, \
                       acc / t * 1e-6, acc / (t * 1e-3) / 256 / 2.4e9);                                                   \
            }
            RUN(0, 1) RUN(0, 4) RUN(1, 1) RUN(1, 4) RUN(1, 3)
#undef RUN
        }
    }
    return 0;
}
