// drt_sincos.h -- the f32 transcendentals of BxDF sampling, written for the hardware's 1-ulp v_exp_f32 / v_log_f32
// instead of libm's multi-branch expf / logf / log1pf / expm1f (30-60 VALU each), host + device so that
// tests/cpp/sincos_kat.cpp can sweep them against libm in double.
//
// (1) sin and cos of phi = 2 pi u for the 31-bit draw r, u = r / RAND_MAX (bxdf.hpp:73,110).
//
// The reduction is done on the INTEGER -- quadrant q = round(4 r / 2^31), remainder xi = r - q 2^29 in
// [-2^28, 2^28], x = xi 2 pi / 2^31 in [-pi/4, pi/4] -- so no precision is lost before the polynomials (the float u
// carries 24 of the draw's 31 bits, this keeps 29) and none of sincospif's general range reduction and special cases
// is executed: 24 instead of 38 VALU on gfx950.  2^31 stands for RAND_MAX = 2^31 - 1: an angle error of 3e-9 rad, 20x
// below f32 resolution.  Cephes' minimax polynomials for |x| <= pi/4.  Host + device: tests/cpp/sincos_kat.cpp sweeps
// the whole range against libm in double (max abs error 1.2e-7).
// (Round 4 measured the hardware's v_sin_f32 / v_cos_f32 on u = r / 2^31 in its place -- arguments in revolutions, the range
// reduction done by the instruction: 8 instead of 24 issue slots, k_path 0.705 -> 0.670 ms on config 3 -- and did not keep it:
// max abs error 2.7e-7 (mean 7e-8) against 1.2e-7 (2.6e-8), tools/microbench_sincos.hip, and with it two of the small
// fixed-seed fixtures each gained a path whose hit decision flips under f32 rounding -- g4b's gradient went to 1.36e-4 of
// its largest component, over the stated 1e-4.  At full size nothing moves (config 3: 4.5e-6), but the small fixtures are
// what pins the f32 mode path for path.  The hybrid -- this integer quadrant reduction, then the two hardware instructions on
// the remainder in place of the two polynomials -- is as accurate as the polynomials (1.29e-7 / mean 4.2e-8) and passes every
// test, but buys 0.4 %: 0.700 -> 0.697 ms.  The quarter-rate pipe is the busier one.)
#pragma once

#if !defined(__HIPCC_RTC__)
#include <stdint.h>
#endif

#if defined(__HIPCC__)
#define DRT_SC_HD __host__ __device__ inline
#else
#define DRT_SC_HD inline
#endif

DRT_SC_HD float drt_bits_to_float(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
DRT_SC_HD uint32_t drt_float_to_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

DRT_SC_HD void sincos_2pi_u31(uint32_t r, float* s, float* c)
{
    const uint32_t q = (r + 0x10000000u) >> 29;                    // 0..4
    const int32_t xi = (int32_t)(r - (q << 29));
    const float x = (float)xi * 2.9258361585343192e-09f;           // 2 pi / 2^31
    const float z = x * x;
    const float sp = x + x * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float cp = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const bool swap = (q & 1u) != 0;
    const float ss = swap ? cp : sp, cc = swap ? sp : cp;
    // quadrant 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s); 4 = 0
    *s = drt_bits_to_float(drt_float_to_bits(ss) ^ ((q & 2u) << 30));
    *c = drt_bits_to_float(drt_float_to_bits(cc) ^ (((q + 1u) & 2u) << 30));
}

// ... the same in f64 (the verification mode; DRT_RENDER_F64): quadrant in quarter turns, Cephes' sin / cos kernels for doubles
DRT_SC_HD void drt_sincos_2pi_u31_f64(uint32_t r, double* s, double* c)
{
    const double x = (double)r * (4.0 / 2147483647.0);            // quarter turns, [0, 4]
    const double qf = __builtin_rint(x);
    const double y = (x - qf) * 1.5707963267948966192;            // [-pi/4, pi/4]
    const uint32_t q = (uint32_t)(int)qf;                         // 0 .. 4
    const double z = y * y;
    const double ps = ((((1.58962301576546568060e-10 * z - 2.50507477628578072866e-8) * z + 2.75573136213857245213e-6) * z
                        - 1.98412698295895385996e-4) * z + 8.33333333332211858878e-3) * z - 1.66666666666666307295e-1;
    const double pc = ((((-1.13585365213876817300e-11 * z + 2.08757008419747316778e-9) * z - 2.75573141792967388112e-7) * z
                        + 2.48015872888517045348e-5) * z - 1.38888888888730564116e-3) * z + 4.16666666666665929218e-2;
    const double sp = y + y * z * ps;
    const double cp = 1.0 - 0.5 * z + z * z * pc;
    const bool swap = (q & 1u) != 0;
    const double ss = swap ? cp : sp, cc = swap ? sp : cp;
    // quadrant 0: (s, c); 1: (c, -s); 2: (-s, -c); 3: (-c, s); 4 = 0
    *s = (q & 2u) ? -ss : ss;
    *c = ((q + 1u) & 2u) ? -cc : cc;
}

// (2) The specular lobe's theta (bxdf.hpp:106-113): cos^2 = u^(2/(e+2)) = exp(x), sin^2 = 1 - exp(x), x = log(u) 2/(e+2).
// log(u) for u = r / RAND_MAX: near u = 1 it is formed from the EXACT integer w = (RAND_MAX - r) / RAND_MAX as the
// series of log1p(-w) (six terms, relative error < 3e-9 for w < 1/16) -- the float u would have lost w's low bits --
// elsewhere as ln2 * v_log_f32(u).  1 - exp(x) likewise: the series of -expm1(x) for |x| < 1/16, 1 - exp2 elsewhere.
#if defined(__HIP_DEVICE_COMPILE__)
#define DRT_HW_LOG2(x) __builtin_amdgcn_logf(x)
#define DRT_HW_EXP2(x) __builtin_amdgcn_exp2f(x)
#else
#include <math.h>
#define DRT_HW_LOG2(x) log2f(x)
#define DRT_HW_EXP2(x) exp2f(x)
#endif

DRT_SC_HD float drt_log_u31(uint32_t r)              // log(r / RAND_MAX), 1 <= r <= RAND_MAX - 1
{
    const float w = (float)(2147483647u - r) * 4.656612875245797e-10f;
    if (w < 0.0625f)
        return -w * (1.f + w * (0.5f + w * (0.33333334f + w * (0.25f + w * (0.2f + w * 0.16666667f)))));
    return 0.6931471805599453f * DRT_HW_LOG2((float)r * 4.656612875245797e-10f);
}

DRT_SC_HD float drt_exp_nonpos(float x)              // exp(x), x <= 0
{
    return DRT_HW_EXP2(x * 1.4426950408889634f);
}

DRT_SC_HD float drt_one_minus_exp(float x)           // 1 - exp(x), x <= 0, to full relative accuracy
{
    if (x > -0.0625f)
        return -x * (1.f + x * (0.5f + x * (0.16666667f + x * (0.041666668f + x * 0.008333334f))));
    return 1.f - DRT_HW_EXP2(x * 1.4426950408889634f);
}
