// drt_sincos.h -- sin and cos of phi = 2 pi u for the 31-bit draw r, u = r / RAND_MAX (bxdf.hpp:73,110), in f32.
//
// The reduction is done on the INTEGER -- quadrant q = round(4 r / 2^31), remainder xi = r - q 2^29 in
// [-2^28, 2^28], x = xi 2 pi / 2^31 in [-pi/4, pi/4] -- so no precision is lost before the polynomials (the float u
// carries 24 of the draw's 31 bits, this keeps 29) and none of sincospif's general range reduction and special cases
// is executed: 24 instead of 38 VALU on gfx950.  2^31 stands for RAND_MAX = 2^31 - 1: an angle error of 3e-9 rad, 20x
// below f32 resolution.  Cephes' minimax polynomials for |x| <= pi/4.  Host + device: tests/cpp/sincos_kat.cpp sweeps
// the whole range against libm in double (max abs error 1.2e-7).
#pragma once

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DRT_SC_HD __host__ __device__ inline
#else
#define DRT_SC_HD inline
#endif

DRT_SC_HD float drt_bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
DRT_SC_HD uint32_t drt_float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

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
