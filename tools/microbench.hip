// microbench.hip -- measured roofs and K2 ablations on the GPU box (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idifferentiable-renderer_amd/csrc tools/microbench.hip -o gpurun_out/microbench
#include "drt_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_copy4(const float4* __restrict__ in, float4* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_read4(const float4* __restrict__ in, float* __restrict__ out, size_t n)
{
    float acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write4(float4* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// K2's memory pattern without its arithmetic: 16 B + 8 B in, 8 B out per element
__global__ void __launch_bounds__(256) k_k2shape(const float4* __restrict__ a, const float2* __restrict__ b, float2* __restrict__ h, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float4 x = a[i]; float2 y = b[i];
        h[i] = make_float2(x.x + x.y + x.z + x.w, y.x + y.y);
    }
}

template <typename F> float time_ms(F f, int reps)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv)
{
    size_t N = argc > 1 ? atoll(argv[1]) : (size_t)1 << 24;
    float4 *a, *c; float2 *b, *h; float* sink;
    CK(hipMalloc(&a, N * 16)); CK(hipMalloc(&c, N * 16)); CK(hipMalloc(&b, N * 8)); CK(hipMalloc(&h, N * 8)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(a, 0, N * 16)); CK(hipMemset(b, 0, N * 8));
    for (int grid : {2048, 4096, 8192, 65536}) {
        float t;
        t = time_ms([&] { hipLaunchKernelGGL(k_copy4, dim3(grid), dim3(256), 0, 0, a, c, N); }, 20);
        printf("grid %6d copy4   %8.3f ms  %7.1f GB/s (r+w)\n", grid, t, 2 * N * 16 / t * 1e-6);
        t = time_ms([&] { hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, a, sink, N); }, 20);
        printf("grid %6d read4   %8.3f ms  %7.1f GB/s\n", grid, t, N * 16 / t * 1e-6);
        t = time_ms([&] { hipLaunchKernelGGL(k_write4, dim3(grid), dim3(256), 0, 0, c, N); }, 20);
        printf("grid %6d write4  %8.3f ms  %7.1f GB/s\n", grid, t, N * 16 / t * 1e-6);
        t = time_ms([&] { hipLaunchKernelGGL(k_k2shape, dim3(grid), dim3(256), 0, 0, a, b, h, N); }, 20);
        printf("grid %6d k2shape %8.3f ms  %7.1f GB/s (32 B/elem)\n", grid, t, N * 32 / t * 1e-6);
    }
    // the real K2 on a synthetic full queue: rays from the origin into the Cornell box
    DevScene<float> hs; memset(&hs, 0, sizeof hs);
    auto plane = [&](int i, float nx, float ny, float nz, float off) { hs.shapes[i].type = 0; hs.shapes[i].p[0] = nx; hs.shapes[i].p[1] = ny; hs.shapes[i].p[2] = nz; hs.shapes[i].p[3] = off; };
    auto sphere = [&](int i, float x, float y, float z, float r) { hs.shapes[i].type = 1; hs.shapes[i].p[0] = x; hs.shapes[i].p[1] = y; hs.shapes[i].p[2] = z; hs.shapes[i].p[3] = r; };
    sphere(0, 0, 0, 3, 1); sphere(1, -1, 1, 4.5f, 1); plane(2, -1, 0, 0, -3); plane(3, 1, 0, 0.1f, -3); plane(4, 0, 0, -1, -6);
    plane(5, 0, 0, 1, 0); plane(6, 0, 1, 0, -3); plane(7, 0, -1, 0, -3); sphere(8, 0, 3, 3, 1);
    std::vector<float4> ha(N); std::vector<float2> hb(N);
    for (size_t i = 0; i < N; ++i) {
        float u = (float)((i * 2654435761u) & 0xFFFF) / 65536.f - 0.5f, v = (float)((i * 40503u) & 0xFFFF) / 65536.f - 0.5f;
        float dx = u, dy = v, dz = 1.f, inv = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
        ha[i] = make_float4(0.f, 0.f, 0.5f, dx * inv); hb[i] = make_float2(dy * inv, dz * inv);
    }
    CK(hipMemcpy(a, ha.data(), N * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), N * 8, hipMemcpyHostToDevice));
    DevScene<float>* ds; CK(hipMalloc(&ds, sizeof hs));
    for (int ns : {0, 1, 9}) {
        hs.n_shapes = ns; CK(hipMemcpy(ds, &hs, sizeof hs, hipMemcpyHostToDevice));
        for (int shift : {8, 10}) {
            BatchArgs ba; memset(&ba, 0, sizeof ba);
            ba.n_paths = (uint32_t)N; ba.region_shift = shift; ba.region_size = 1u << shift; ba.n_regions = (uint32_t)(N >> shift);
            std::vector<uint32_t> hc(ba.n_regions, ba.region_size);
            uint32_t* dc; CK(hipMalloc(&dc, hc.size() * 4)); CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
            for (int grid : {1024, 2048, 4096}) {
                float t = time_ms([&] { hipLaunchKernelGGL(k_intersect<float>, dim3(grid), dim3(256), 0, 0, ba, ds, a, b, (HitRec<float>*)h, dc, (unsigned long long*)nullptr); }, 20);
                printf("k_intersect shapes %d region %4u grid %5d: %8.3f ms  %7.1f GB/s  %6.1f Gray/s\n", ns, ba.region_size, grid, t, N * 32 / t * 1e-6, N / t * 1e-6);
            }
            CK(hipFree(dc));
        }
    }
    return 0;
}
