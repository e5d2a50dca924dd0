// Known-answer sweep of the device's integer-reduced sin/cos (csrc/drt_sincos.h, compiled here for the host):
// every 1021st draw of the 31-bit range plus both ends densely, against libm in double.
#include "../../differentiable-renderer_amd/csrc/drt_sincos.h"

#include <cmath>
#include <cstdio>
#include <initializer_list>

int main()
{
    double worst = 0;
    uint32_t at = 0;
    for (uint64_t r = 0; r <= 2147483647ull; r += (r < 100000 || r > 2147383647ull) ? 1 : 1021) {
        float s, c;
        sincos_2pi_u31((uint32_t)r, &s, &c);
        const double phi = 2 * M_PI * (double)r / 2147483647.0;
        const double e = std::fmax(std::fabs(s - std::sin(phi)), std::fabs(c - std::cos(phi)));
        if (e > worst) { worst = e; at = (uint32_t)r; }
    }
    float s, c;
    sincos_2pi_u31(0u, &s, &c);
    const bool ends = s == 0.f && c == 1.f;
    sincos_2pi_u31(1073741824u, &s, &c);                 // half a turn: (0, -1)
    const bool half = std::fabs(s) < 1e-7f && c == -1.f;
    std::printf("sin/cos: max abs error %.3e at r = %u\n", worst, at);
    // the f64 mode's sin / cos (quarter-turn reduction + Cephes' double kernels) against libm, in long double
    long double worst64 = 0;
    for (uint64_t r = 0; r <= 2147483647ull; r += (r < 100000 || r > 2147383647ull) ? 1 : 509) {
        double s64, c64;
        drt_sincos_2pi_u31_f64((uint32_t)r, &s64, &c64);
        const long double phi = 2 * 3.14159265358979323846264338327950288L * (long double)r / 2147483647.0L;
        const long double e = fmaxl(fabsl((long double)s64 - sinl(phi)), fabsl((long double)c64 - cosl(phi)));
        if (e > worst64) worst64 = e;
    }
    double s0, c0, sq, cq;
    drt_sincos_2pi_u31_f64(0u, &s0, &c0);
    drt_sincos_2pi_u31_f64(2147483647u, &sq, &cq);
    std::printf("sin/cos f64: max abs error %.3Le\n", worst64);
    if (!(worst64 < 2e-15L) || s0 != 0.0 || c0 != 1.0 || std::fabs(sq) > 1e-15 || cq != 1.0) { std::printf("f64 sin/cos FAILED\n"); return 1; }
    // the specular lobe's helpers: log(u), exp(x), 1 - exp(x) against libm in double (relative errors)
    double e_log = 0, e_exp = 0, e_ome = 0;
    for (uint64_t r = 1; r < 2147483647ull; r += (r < 200000 || r > 2147283647ull) ? 1 : 4099) {
        const double lu = std::log((double)r / 2147483647.0);
        e_log = std::fmax(e_log, std::fabs(drt_log_u31((uint32_t)r) - lu) / std::fabs(lu));
        for (double e : {1.0, 30.0, 80.0}) {
            const float x = (float)(lu * 2.0 / (e + 2.0));
            e_exp = std::fmax(e_exp, std::fabs(drt_exp_nonpos(x) - std::exp((double)x)) / std::exp((double)x));
            const double ome = -std::expm1((double)x);
            if (ome > 0)
                e_ome = std::fmax(e_ome, std::fabs(drt_one_minus_exp(x) - ome) / ome);
        }
    }
    std::printf("log(u): max rel error %.3e; exp(x): %.3e; 1 - exp(x): %.3e\n", e_log, e_exp, e_ome);
    if (worst < 2e-7 && ends && half && e_log < 2e-6 && e_exp < 2e-6 && e_ome < 3e-6) { std::printf("ok\n"); return 0; }
    return 1;
}
