// Known-answer sweep of the device's integer-reduced sin/cos (csrc/drt_sincos.h, compiled here for the host):
// every 1021st draw of the 31-bit range plus both ends densely, against libm in double.
#include "../../differentiable-renderer_amd/csrc/drt_sincos.h"

#include <cmath>
#include <cstdio>

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
    std::printf("max abs error %.3e at r = %u\n", worst, at);
    if (worst < 2e-7 && ends && half) { std::printf("ok\n"); return 0; }
    return 1;
}
