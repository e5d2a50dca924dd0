// microbench_imul.hip -- issue cost of the integer multiplies of the draw hash on gfx950 (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_imul.hip -o gpurun_out/microbench_imul
// Every kernel runs N dependent-chain-free rounds of ONE instruction kind on 8 independent accumulators per lane, 5 waves per
// SIMD resident: time per wave-instruction = what the instruction occupies the SIMD's issue port for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { OP_ADD = 0, OP_MUL_LO = 1, OP_MUL_U24 = 2, OP_MAD_U24 = 3, OP_XOR_SHIFT = 4, OP_FMA = 5, OP_MUL_HI = 6, OP_MAD_U64 = 7, OP_RCP = 8, OP_SQRT = 9 };

// (inline asm: the compiler folds chains of x + c or x * c into one instruction)
#define ASM3(name) asm volatile(name " %0, %1, %2" : "=v"(r) : "v"(x), "v"(c))
template <int OP>
__device__ inline uint32_t step(uint32_t x, uint32_t c)
{
    uint32_t r = x;
    if (OP == OP_ADD) ASM3("v_add_u32");
    if (OP == OP_MUL_LO) ASM3("v_mul_lo_u32");
    if (OP == OP_MUL_U24) ASM3("v_mul_u32_u24");
    if (OP == OP_MAD_U24) asm volatile("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_XOR_SHIFT) { asm volatile("v_lshrrev_b32 %0, 15, %1" : "=v"(r) : "v"(x)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(x)); }
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_MUL_HI) ASM3("v_mul_hi_u32");
    if (OP == OP_MAD_U64) { uint64_t w = x; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w) : "v"(x), "v"(c) : "vcc"); r = (uint32_t)w; }
    if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %1" : "=v"(r) : "v"(x));
    if (OP == OP_SQRT) asm volatile("v_sqrt_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int OP>
__global__ void __launch_bounds__(256) k_ops(uint32_t* out, uint32_t c, int rounds)
{
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a[i] = threadIdx.x * 2654435761u + i * 40503u + c;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                a[i] = step<OP>(a[i], c);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s ^= a[i];
    if (s == 0x12345678u)
        out[0] = s;
}

template <int OP>
void run(const char* name, uint32_t* out, int per_step)
{
    const int rounds = 4096, grid = 256 * 5;     // 5 blocks of 4 waves per CU: 5 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_ops<OP>, dim3(grid), dim3(256), 0, 0, out, 0x7feb352du, rounds);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(k_ops<OP>, dim3(grid), dim3(256), 0, 0, out, 0x7feb352du, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_instr = (double)grid * 4 * rounds * 32 * per_step;
    printf("%-12s %8.3f ms  %8.1f G wave-instr/s  (%.2f of 1228.8)\n", name, ms, wave_instr / ms * 1e-6, wave_instr / ms * 1e-6 / 1228.8);
}

int main()
{
    uint32_t* out; CK(hipMalloc(&out, 64));
    run<OP_ADD>("v_add_u32", out, 1);
    run<OP_FMA>("v_fma_f32", out, 1);
    run<OP_MUL_LO>("v_mul_lo_u32", out, 1);
    run<OP_MUL_HI>("v_mul_hi_u32", out, 1);
    run<OP_MUL_U24>("v_mul_u32_u24", out, 1);
    run<OP_MAD_U24>("v_mad_u32_u24", out, 1);
    run<OP_MAD_U64>("mad_u64_u32", out, 1);
    run<OP_XOR_SHIFT>("shift+xor", out, 2);
    run<OP_RCP>("v_rcp_f32", out, 1);
    run<OP_SQRT>("v_sqrt_f32", out, 1);
    return 0;
}
