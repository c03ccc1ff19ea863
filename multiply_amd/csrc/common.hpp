// Small device helpers shared by the sampler, warp and compositing kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace mp {

// LaplaceDensity.density_func (reference code/lib/model/density.py:20-29):
//   sigma = (1/beta) * (0.5 + 0.5 * sign(sdf) * expm1(-|sdf| / beta))
__device__ __forceinline__ float laplace_density(float sdf, float beta) {
    const float sgn = sdf > 0.0f ? 1.0f : (sdf < 0.0f ? -1.0f : 0.0f);
    return (1.0f / beta) * (0.5f + 0.5f * sgn * expm1f(-fabsf(sdf) / beta));
}

// alpha of one sample exactly as the compositing kernel computes it (multiply.py:455 via nerfacc):
//   alpha = 1 - exp(-sigma * dt)
__device__ __forceinline__ float alpha_of(float sdf, float beta, float dt) {
    return 1.0f - expf(-(laplace_density(sdf, beta) * dt));
}

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// exclusive prefix sum across the 64 lanes of a wave; `total` receives the wave sum
__device__ __forceinline__ float wave_excl_scan(float v, float& total) {
    float incl = v;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(incl, o);
        if (lane >= o) incl += n;
    }
    total = __shfl(incl, 63);
    return incl - v;
}

}  // namespace mp
