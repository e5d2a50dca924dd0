// microbench_imul.hip -- issue cost of the integer multiplies of the draw hash on gfx950 (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_imul.hip -o gpurun_out/microbench_imul
// Every kernel runs N dependent-chain-free rounds of ONE instruction kind on 8 independent accumulators per lane, 5 waves per
// SIMD resident: time per wave-instruction = what the instruction occupies the SIMD's issue port for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { OP_ADD = 0, OP_MUL_LO = 1, OP_MUL_U24 = 2, OP_MAD_U24 = 3, OP_XOR_SHIFT = 4, OP_FMA = 5, OP_MUL_HI = 6, OP_MAD_U64 = 7, OP_RCP = 8, OP_SQRT = 9, OP_PK_FMA = 10, OP_PK_MUL = 11, OP_MAX3 = 12, OP_CVT_UBYTE = 13, OP_CNDMASK = 14, OP_CNDMASK_SGPR = 15, OP_MAX = 16, OP_MULF = 17, OP_ADDF = 18, OP_AND = 19, OP_BFE = 20, OP_CVT_U32 = 21, OP_CMP_CND = 22, OP_MOV = 23, OP_FMAC = 24, OP_MED3 = 25, OP_RSQ = 26, OP_CMP = 27, OP_LSHL_ADD = 28, OP_FMA_K = 29, OP_PERM = 30, OP_FMA_MIX = 31, OP_OR_SDWA = 32, OP_MIN_U32 = 33, OP_CMP_U32 = 34, OP_BFI = 35, OP_CVT_SDWA = 36, OP_MOV_DPP = 37, OP_ADD_DPP = 38, OP_MIN3_U32 = 39, OP_AND_OR = 40, OP_CVT_F16 = 41, OP_MAXIMUM3 = 42, OP_ALIGNBIT = 43, OP_ADD3 = 44, OP_OR3 = 45, OP_SUB_U32 = 46, OP_LSHLREV = 47, OP_ASHR = 48, OP_CMPX = 49 };

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
    if (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(r) : "v"(x));
    if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_MAX) ASM3("v_max_f32");
    if (OP == OP_MULF) ASM3("v_mul_f32");
    if (OP == OP_ADDF) ASM3("v_add_f32");
    if (OP == OP_AND) ASM3("v_and_b32");
    if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(r) : "v"(x));
    if (OP == OP_CVT_U32) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(r) : "v"(x));
    if (OP == OP_CMP_CND) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x), "v"(c) : "vcc"); asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r) : "v"(x), "v"(c) : "vcc"); }
    if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
    if (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(x), "v"(c));
    if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_RSQ) asm volatile("v_rsq_f32 %0, %1" : "=v"(r) : "v"(x));
    if (OP == OP_CMP) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x), "v"(c) : "vcc"); }
    if (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %1, %2, %2" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %1 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_OR_SDWA) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_MIN_U32) ASM3("v_min_u32");
    if (OP == OP_CMP_U32) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" :: "v"(x), "v"(c) : "vcc"); }
    if (OP == OP_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_CVT_SDWA) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(x));
    if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    if (OP == OP_ADD_DPP) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_MIN3_U32) asm volatile("v_min3_u32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_AND_OR) asm volatile("v_and_or_b32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_CVT_F16) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(r) : "v"(x));
    if (OP == OP_MAXIMUM3) asm volatile("v_maximum3_f32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_ALIGNBIT) asm volatile("v_alignbit_b32 %0, %1, %2, 8" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_OR3) asm volatile("v_or3_b32 %0, %1, %2, %1" : "=v"(r) : "v"(x), "v"(c));
    if (OP == OP_SUB_U32) ASM3("v_sub_u32");
    if (OP == OP_LSHLREV) asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(r) : "v"(x));
    if (OP == OP_ASHR) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(x));
    if (OP == OP_CMPX) { asm volatile("v_cmpx_lt_f32 exec, %0, %1\n\ts_mov_b64 exec, -1" :: "v"(x), "v"(c) : "exec"); }
    if (OP == OP_FMA_K) asm volatile("v_fmaak_f32 %0, %1, %1, 0x3f7feb35" : "=v"(r) : "v"(x));
    return r;
}

// packed f32: two values per lane and instruction (the accumulators are register pairs)
template <int OP>
__global__ void __launch_bounds__(256) k_pk(uint32_t* out, uint32_t c, int rounds)
{
    uint64_t a[8];
    const uint64_t cc = ((uint64_t)c << 32) | c;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a[i] = (uint64_t)(threadIdx.x * 2654435761u + i * 40503u + c) * 0x100000001ull;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(a[i]) : "v"(a[i]), "v"(cc));
                if (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(cc));
            }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s ^= a[i];
    if (s == 0x12345678u)
        out[0] = (uint32_t)s;
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
void run_pk(const char* name, uint32_t* out)
{
    const int rounds = 4096, grid = 256 * 5;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_pk<OP>, dim3(grid), dim3(256), 0, 0, out, 0x3f7feb35u, rounds);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(k_pk<OP>, dim3(grid), dim3(256), 0, 0, out, 0x3f7feb35u, rounds);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double wave_instr = (double)grid * 4 * rounds * 32;
    printf("%-26s %8.3f ms  %8.1f G wave-instr/s  (%.2f of 1228.8; two f32 results per lane each)\n", name, ms, wave_instr / ms * 1e-6, wave_instr / ms * 1e-6 / 1228.8);
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
    printf("%-26s %8.3f ms  %8.1f G wave-instr/s  (%.2f of 1228.8)\n", name, ms, wave_instr / ms * 1e-6, wave_instr / ms * 1e-6 / 1228.8);
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
    run<OP_MAX3>("v_max3_f32", out, 1);
    run<OP_CVT_UBYTE>("v_cvt_f32_ubyte1", out, 1);
    run<OP_CNDMASK>("v_cndmask_b32", out, 1);
    run<OP_CNDMASK_SGPR>("v_cndmask_b32 (sgpr pair)", out, 1);
    run<OP_CMP>("v_cmp_lt_f32", out, 1);
    run<OP_CMP_CND>("v_cmp + v_cndmask", out, 2);
    run<OP_MAX>("v_max_f32", out, 1);
    run<OP_MED3>("v_med3_f32", out, 1);
    run<OP_MULF>("v_mul_f32", out, 1);
    run<OP_ADDF>("v_add_f32", out, 1);
    run<OP_FMAC>("v_fmac_f32", out, 1);
    run<OP_FMA_K>("v_fma_f32 with a literal", out, 1);
    run<OP_AND>("v_and_b32", out, 1);
    run<OP_BFE>("v_bfe_u32", out, 1);
    run<OP_LSHL_ADD>("v_lshl_add_u32", out, 1);
    run<OP_CVT_U32>("v_cvt_f32_u32", out, 1);
    run<OP_MOV>("v_mov_b32", out, 1);
    run<OP_RSQ>("v_rsq_f32", out, 1);
    run<OP_PERM>("v_perm_b32", out, 1);
    run<OP_FMA_MIX>("v_fma_mix_f32 (f16 src0)", out, 1);
    run<OP_OR_SDWA>("v_or_b32_sdwa byte sel", out, 1);
    run<OP_CVT_SDWA>("v_cvt_f32_u32_sdwa byte", out, 1);
    run<OP_CVT_F16>("v_cvt_f32_f16", out, 1);
    run<OP_MIN_U32>("v_min_u32", out, 1);
    run<OP_MIN3_U32>("v_min3_u32", out, 1);
    run<OP_MAXIMUM3>("v_maximum3_f32", out, 1);
    run<OP_CMP_U32>("v_cmp_lt_u32", out, 1);
    run<OP_CMPX>("v_cmpx_lt_f32 (+ s_mov exec)", out, 1);
    run<OP_BFI>("v_bfi_b32", out, 1);
    run<OP_AND_OR>("v_and_or_b32", out, 1);
    run<OP_OR3>("v_or3_b32", out, 1);
    run<OP_ADD3>("v_add3_u32", out, 1);
    run<OP_SUB_U32>("v_sub_u32", out, 1);
    run<OP_ALIGNBIT>("v_alignbit_b32", out, 1);
    run<OP_LSHLREV>("v_lshlrev_b32", out, 1);
    run<OP_ASHR>("v_ashrrev_i32", out, 1);
    run<OP_MOV_DPP>("v_mov_b32_dpp quad_perm", out, 1);
    run<OP_ADD_DPP>("v_add_f32_dpp quad_perm", out, 1);
    run_pk<OP_PK_FMA>("v_pk_fma_f32", out);
    run_pk<OP_PK_MUL>("v_pk_mul_f32", out);
    return 0;
}
