#include "prelude.h"
#include "drt_path.h"
extern "C" __global__ void __launch_bounds__(DRT_BLOCK)
drt_jit_k_path(PathArgs a, const DevScene<float>* __restrict__ sc, const float* __restrict__ params, const float* __restrict__ adjoint,
       double* __restrict__ gpart, double* __restrict__ fpart, uint32_t* __restrict__ counts,
       unsigned long long* __restrict__ total, double* __restrict__ gimg_part);
template __global__ void k_path<float, false, 4, 3, DRT_SIG_CORNELL, DRT_NSIG_CORNELL, false>(PathArgs, const DevScene<float>*, const float*, const float*, double*, double*, uint32_t*, unsigned long long*, double*);
