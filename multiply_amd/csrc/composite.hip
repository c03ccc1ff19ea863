// Multi-person compositing (reference code/lib/model/multiply.py:425-480, 544-545, 590).
// The reference packs every person's samples into one list, radix-sorts it twice (by t_end, then stably by ray) and
// calls nerfacc's packed scan; the arithmetic per sample is nerfacc's render_weight_from_density:
//   alpha = 1 - exp(-sigma dt),  T = exp(-sum_{earlier samples of the ray} sigma dt),  w = alpha T.
// Here ONE WAVE owns one ray.  Each person's samples are already sorted, so "earlier in the merged order" needs no
// merge: the free energy in front of sample (p,i) is person p's own exclusive prefix sum plus, for every other person q,
// q's prefix sum at the rank of t_end(p,i) among q's t_ends (binary search; ties: lower person first, like the stable
// sorts).  Rows are read with coalesced 64-lane loads (the first version, one thread per ray walking its rows, moved
// 15x the algorithmic bytes through HBM: profiles/r01_pmc_traffic.txt), prefix sums are wave scans.
#include <hip/hip_runtime.h>
#include <float.h>
#include "../../include/multiply_hip.h"
#include "common.hpp"

namespace {

constexpr int MAX_P = 8;

constexpr int WPB = 4;   // rays (waves) per workgroup

__global__ __launch_bounds__(64 * WPB) void k_composite(int n_rays, int P, int n_z, const int* const* __restrict__ inv_index,
                                                       const float* const* __restrict__ z,
                                                       const float* const* __restrict__ sdf,
                                                       const float* const* __restrict__ rgb,
                                                       const float* const* __restrict__ normal,
                                                       const float* __restrict__ beta_p, const float* __restrict__ bg_rgb,
                                                       float* __restrict__ rgb_values, float* __restrict__ fg_rgb_values,
                                                       float* __restrict__ normal_values, float* __restrict__ acc_map,
                                                       float* __restrict__ acc_person, float* __restrict__ bg_T) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * WPB + wave;
    const int S = n_z - 1;
    // per wave, per person: te [S] (t_end), fe [S] (free energy sigma dt), pf [S+1] (prefix sums, pf[0] = 0)
    const int per_person = 3 * S + 1;
    float* base = smem + (size_t)wave * P * per_person;
    const float beta = *beta_p;
    const bool live = r < n_rays;
    int k[MAX_P];
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) k[p] = (live && p < P) ? inv_index[p][r] : -1;

    // ---- phase 1: free energies and per-person prefix sums
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) {
        if (p < P && k[p] >= 0) {
            float* te_l = base + p * per_person;
            float* fe_l = te_l + S;
            float* pf_l = fe_l + S;
            const float* zr = z[p] + (size_t)k[p] * n_z;
            const float* sr = sdf[p] + (size_t)k[p] * S;
            float carry = 0.f;
            if (lane == 0) pf_l[0] = 0.f;
            for (int i0 = 0; i0 < S; i0 += 64) {
                const int i = i0 + lane;
                float fe = 0.f;
                if (i < S) {
                    const float ts = zr[i], te = zr[i + 1];
                    fe = mp::laplace_density(sr[i], beta) * (te - ts);
                    te_l[i] = te;
                    fe_l[i] = fe;
                }
                float tot;
                const float ex = mp::wave_excl_scan(fe, tot);
                if (i < S) pf_l[i + 1] = carry + ex + fe;
                carry += tot;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: weights and accumulation
    float c[3] = {0.f, 0.f, 0.f}, nn[3] = {0.f, 0.f, 0.f}, acc = 0.f, accp[MAX_P];
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) accp[p] = 0.f;
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) {
        if (p < P && k[p] >= 0) {
            const float* te_l = base + p * per_person;
            const float* fe_l = te_l + S;
            const float* pf_l = fe_l + S;
            for (int i = lane; i < S; i += 64) {
                const float te = te_l[i], fe = fe_l[i];
                float E = pf_l[i];
#pragma unroll
                for (int q = 0; q < MAX_P; ++q) {
                    if (q < P && q != p && k[q] >= 0) {
                        const float* te_q = base + q * per_person;
                        // number of q's samples in front: t_end < te (q > p) or <= te (q < p)
                        int lo = 0, hi = S;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            const float t = te_q[mid];
                            const bool before = q < p ? t <= te : t < te;
                            if (before) lo = mid + 1; else hi = mid;
                        }
                        E += (te_q + 2 * S)[lo];
                    }
                }
                const float T = expf(-E);
                const float w = (1.0f - expf(-fe)) * T;
                if (w != 0.0f) {  // skipped samples carry undefined colour/normal but exactly zero weight
                    const size_t qi = (size_t)k[p] * S + i;
                    c[0] += w * rgb[p][3 * qi]; c[1] += w * rgb[p][3 * qi + 1]; c[2] += w * rgb[p][3 * qi + 2];
                    nn[0] += w * normal[p][3 * qi]; nn[1] += w * normal[p][3 * qi + 1]; nn[2] += w * normal[p][3 * qi + 2];
                }
                acc += w;
                accp[p] += w;
            }
        }
    }
    for (int a = 0; a < 3; ++a) { c[a] = mp::wsum(c[a]); nn[a] = mp::wsum(nn[a]); }
    acc = mp::wsum(acc);
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) accp[p] = mp::wsum(accp[p]);
    if (!live || lane != 0) return;
    // exclusive transmittance of the LAST packed sample (multiply.py:457-463): the last sample of the person whose final
    // t_end is the largest (ties: higher person); 1 for rays without samples
    int p_last = -1;
    float tm = -FLT_MAX;
    for (int p = 0; p < P; ++p)
        if (k[p] >= 0 && (base + p * per_person)[S - 1] >= tm) { tm = (base + p * per_person)[S - 1]; p_last = p; }
    const bool any = p_last >= 0;
    float E_front = 0.f;   // everything in front of the last sample = all of the other persons + the last person's first S-1
    for (int p = 0; p < P; ++p)
        if (k[p] >= 0) E_front += (base + p * per_person + 2 * S)[p == p_last ? S - 1 : S];
    const float T_last = any ? expf(-E_front) : 1.0f;
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
    const int S = n_z - 1;
    const int lds = WPB * n_person * (3 * S + 1) * (int)sizeof(float);
    if (lds > 160 * 1024) return -2;
    MP_LDS_ATTR((k_composite), 160 * 1024);
    hipLaunchKernelGGL(k_composite, dim3((n_rays + WPB - 1) / WPB), dim3(64 * WPB), lds, (hipStream_t)stream, n_rays, n_person, n_z,
                       inv_index, z, sdf, rgb, normal, beta, bg_rgb, rgb_values, fg_rgb_values, normal_values, acc_map,
                       acc_person, bg_T);
    return (int)hipGetLastError();
}
