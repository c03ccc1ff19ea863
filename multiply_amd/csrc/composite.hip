// Multi-person compositing (reference code/lib/model/multiply.py:425-480, 544-545, 590).
// The reference packs every person's samples into one list, radix-sorts it twice (by t_end, then stably by ray) and
// calls nerfacc's packed scan.  Here one thread owns one ray and merges the (already sorted) per-person lists on the
// fly; the arithmetic per sample is nerfacc's render_weight_from_density:
//   alpha = 1 - exp(-sigma dt),  T = exp(-sum_{earlier samples of the ray} sigma dt),  w = alpha T.
#include <hip/hip_runtime.h>
#include <float.h>
#include "../../include/multiply_hip.h"
#include "common.hpp"

namespace {

constexpr int MAX_P = 8;

__global__ __launch_bounds__(256) void k_composite(int n_rays, int P, int n_z, const int* const* __restrict__ inv_index,
                                                   const float* const* __restrict__ z,
                                                   const float* const* __restrict__ sdf,
                                                   const float* const* __restrict__ rgb,
                                                   const float* const* __restrict__ normal,
                                                   const float* __restrict__ beta_p, const float* __restrict__ bg_rgb,
                                                   float* __restrict__ rgb_values, float* __restrict__ fg_rgb_values,
                                                   float* __restrict__ normal_values, float* __restrict__ acc_map,
                                                   float* __restrict__ acc_person, float* __restrict__ bg_T) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int S = n_z - 1;
    const float beta = *beta_p;
    int k[MAX_P], cur[MAX_P];
    float accp[MAX_P];
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) {
        k[p] = p < P ? inv_index[p][r] : -1;
        cur[p] = 0;
        accp[p] = 0.f;
    }
    float csum = 0.f, T_last = 1.0f, c[3] = {0.f, 0.f, 0.f}, nn[3] = {0.f, 0.f, 0.f}, acc = 0.f;
    for (;;) {
        int best = -1;
        float te_best = FLT_MAX;
#pragma unroll
        for (int p = 0; p < MAX_P; ++p) {
            if (k[p] >= 0 && cur[p] < S) {
                const float te = z[p][(size_t)k[p] * n_z + cur[p] + 1];
                if (te < te_best) { te_best = te; best = p; }  // ties: lower person first
            }
        }
        if (best < 0) break;
        const int p = best, i = cur[p];
        const size_t q = (size_t)k[p] * S + i;
        const float ts = z[p][(size_t)k[p] * n_z + i];
        const float fe = mp::laplace_density(sdf[p][q], beta) * (te_best - ts);
        const float alpha = 1.0f - expf(-fe);
        const float T = expf(-csum);
        const float w = alpha * T;
        if (w != 0.0f) {  // skipped samples carry undefined colour/normal but exactly zero weight
            c[0] += w * rgb[p][3 * q]; c[1] += w * rgb[p][3 * q + 1]; c[2] += w * rgb[p][3 * q + 2];
            nn[0] += w * normal[p][3 * q]; nn[1] += w * normal[p][3 * q + 1]; nn[2] += w * normal[p][3 * q + 2];
        }
        acc += w;
        accp[p] += w;
        T_last = T;  // exclusive transmittance of the last packed sample (multiply.py:457-463)
        csum += fe;
        cur[p] = i + 1;
    }
    bg_T[r] = T_last;
    acc_map[r] = acc;
    for (int p = 0; p < P; ++p) acc_person[(size_t)r * P + p] = accp[p];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float bg = bg_rgb ? bg_rgb[3 * r + a] : 1.0f;     // multiply.py:541
        rgb_values[3 * r + a] = c[a] + T_last * bg;             // :544-545
        fg_rgb_values[3 * r + a] = c[a] + T_last * 1.0f;        // :590
        normal_values[3 * r + a] = nn[a];
    }
}

}  // namespace

extern "C" int mp_composite(int n_rays, int n_person, int n_z, const int* const* inv_index, const float* const* z,
                            const float* const* sdf, const float* const* rgb, const float* const* normal,
                            const float* beta, const float* bg_rgb, float* rgb_values, float* fg_rgb_values,
                            float* normal_values, float* acc_map, float* acc_person, float* bg_T, void* stream) {
    if (n_person > MAX_P || n_person < 0) return -1;
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(k_composite, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_rays, n_person, n_z,
                       inv_index, z, sdf, rgb, normal, beta, bg_rgb, rgb_values, fg_rgb_values, normal_values, acc_map,
                       acc_person, bg_T);
    return (int)hipGetLastError();
}
