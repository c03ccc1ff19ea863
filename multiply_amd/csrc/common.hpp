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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: a `static int once = hipFuncSetAttribute(...)` covers
// only the device that is current at the first call.  MP_LDS_ATTR(kernel, bytes) sets it once per (call site, device ordinal)
// and makes the enclosing entry point return the HIP error code if the runtime refuses.
inline int lds_attr(const void* kernel, int bytes, unsigned long long& done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 64 && ((done_mask >> dev) & 1ull)) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev < 64) done_mask |= 1ull << dev;
    return 0;
}
#define MP_LDS_ATTR(kernel, bytes)                                                    \
    do {                                                                              \
        static unsigned long long mp_lds_done_ = 0;                                   \
        const int mp_lds_rc_ = mp::lds_attr((const void*)(kernel), (bytes), mp_lds_done_); \
        if (mp_lds_rc_) return mp_lds_rc_;                                            \
    } while (0)

}  // namespace mp
