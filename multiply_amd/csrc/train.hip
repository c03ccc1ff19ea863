// Element-wise / small kernels of the fp32 TRAINING path (layer-wise forward with stash + hand-written backward).
// Forward-mode convention: a network evaluated for P points in forward mode works on tensors of 4P rows:
//   rows [0,P) values, rows [P,2P) d/dx, [2P,3P) d/dy, [3P,4P) d/dz  (tangents w.r.t. the canonical point x_c).
// The backward pass is the reverse sweep over this forward-mode graph, which yields the mixed second derivatives the
// reference obtains with create_graph=True (normals: multiply.py:620-661, eikonal term: :328-330).
// Entry points: include/multiply_hip.h (mp_tr_*).
#include <hip/hip_runtime.h>
#include <float.h>
#include "../../include/multiply_hip.h"
#include "common.hpp"

namespace {

constexpr int TB = 256;
inline dim3 grid1(long long n) { return dim3((unsigned)((n + TB - 1) / TB)); }

// torch.nn.Softplus(beta=100, threshold=20) and its first two derivatives
__device__ __forceinline__ void softplus_d012(float z, float& h, float& d1, float& d2) {
    const float t = 100.0f * z;
    if (t > 20.0f) { h = z; d1 = 1.0f; d2 = 0.0f; return; }
    const float e = expf(t);
    h = log1pf(e) * 0.01f;
    d1 = e / (1.0f + e);               // sigmoid(100 z)
    d2 = 100.0f * d1 * (1.0f - d1);
}

// ---- Fourier features (embedders.py) and their spatial tangents: out [(FWD?4:1)*P rows][ld], columns col0..col0+NF
template <int D>
__global__ void k_pe_fwd(const float* __restrict__ x, int P, int L, int fwd, float scale, float* __restrict__ out, int ld,
                         int col0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float v[D];
    for (int a = 0; a < D; ++a) v[a] = x[(size_t)i * D + a];
    float* o = out + (size_t)i * ld + col0;
    for (int a = 0; a < D; ++a) o[a] = v[a] * scale;
    for (int k = 0; k < L; ++k) {
        const float f = (float)(1 << k);
        for (int a = 0; a < D; ++a) {
            o[D + 2 * D * k + a] = sinf(v[a] * f) * scale;
            o[D + 2 * D * k + D + a] = cosf(v[a] * f) * scale;
        }
    }
    if (fwd) {
        const int NF = D + 2 * D * L;
        for (int b = 0; b < D; ++b) {
            float* ot = out + (size_t)((b + 1) * P + i) * ld + col0;
            for (int c = 0; c < NF; ++c) ot[c] = 0.0f;
            ot[b] = scale;
            for (int k = 0; k < L; ++k) {
                const float f = (float)(1 << k);
                ot[D + 2 * D * k + b] = f * cosf(v[b] * f) * scale;
                ot[D + 2 * D * k + D + b] = -f * sinf(v[b] * f) * scale;
            }
        }
    }
}

// ---- softplus layer: Z [rows][C] (ldz) -> H [rows][ldh] at col0, times scale; forward mode when P > 0 (rows = 4P)
__global__ void k_softplus_fwd(const float* __restrict__ Z, int ldz, int rows, int C, int P, float scale,
                               float* __restrict__ H, int ldh, int col0) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * C) return;
    const int r = (int)(idx / C), c = (int)(idx % C);
    float h, d1, d2;
    if (P > 0 && r >= P) {
        softplus_d012(Z[(size_t)(r % P) * ldz + c], h, d1, d2);
        H[(size_t)r * ldh + col0 + c] = d1 * Z[(size_t)r * ldz + c] * scale;
    } else {
        softplus_d012(Z[(size_t)r * ldz + c], h, d1, d2);
        H[(size_t)r * ldh + col0 + c] = h * scale;
    }
}

// adjoint: dH (adjoints of the scaled outputs, [rows][ldh] at col0) -> dZ [rows][C]
//   tangent rows: du = s' * dt * scale ; value rows: dz = s' * dh * scale + sum_k s'' * u_k * dt_k * scale
__global__ void k_softplus_bwd(const float* __restrict__ Z, int ldz, int rows, int C, int P, float scale,
                               const float* __restrict__ dH, int ldh, int col0, float* __restrict__ dZ, int lddz) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int vrows = P > 0 ? P : rows;
    if (idx >= (long long)vrows * C) return;
    const int r = (int)(idx / C), c = (int)(idx % C);
    float h, d1, d2;
    softplus_d012(Z[(size_t)r * ldz + c], h, d1, d2);
    float dz = d1 * dH[(size_t)r * ldh + col0 + c] * scale;
    if (P > 0) {
#pragma unroll
        for (int k = 1; k <= 3; ++k) {
            const size_t rk = (size_t)(k * P + r);
            const float dt = dH[rk * ldh + col0 + c] * scale;
            dz += d2 * Z[rk * ldz + c] * dt;
            dZ[rk * lddz + c] = d1 * dt;
        }
    }
    dZ[(size_t)r * lddz + c] = dz;
}

__global__ void k_relu_bwd(const float* __restrict__ H, int ldh, long long n, int C, const float* __restrict__ dH, int lddh,
                           float* __restrict__ dZ, int lddz) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const size_t r = idx / C;
    const int c = (int)(idx % C);
    dZ[r * lddz + c] = H[r * ldh + c] > 0.0f ? dH[r * lddh + c] : 0.0f;
}

// ---- normals + render-net input.  Z8 [4P][257]: col 0 = sdf (value rows) / d sdf (tangent rows), cols 1.. = features
// XA [n][6] = [x_c (3), n (3)] (the colour net reads the 256 features in place from Z8);
// n = normalize(normalize(g . Jinv), eps 1e-6)   (multiply.py:606, 661)
__global__ void k_shade_in_fwd(const float* __restrict__ Z8, int P, int n_pts, const float* __restrict__ xc,
                               const float* __restrict__ jinv, float* __restrict__ XR, float* __restrict__ nrm_out,
                               float* __restrict__ sdf_out, const float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pts) return;
    const int ld = 257;
    // d sdf / d x: the tangent rows of a forward-mode batch, or a separate [P][3] array (reverse-over-reverse net)
    const float g[3] = {grad ? grad[3 * (size_t)i] : Z8[(size_t)(P + i) * ld],
                        grad ? grad[3 * (size_t)i + 1] : Z8[(size_t)(2 * P + i) * ld],
                        grad ? grad[3 * (size_t)i + 2] : Z8[(size_t)(3 * P + i) * ld]};
    const float* J = jinv + 9 * (size_t)i;
    float v[3];
    for (int j = 0; j < 3; ++j) v[j] = g[0] * J[j] + g[1] * J[3 + j] + g[2] * J[6 + j];
    const float a = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
    float n1[3] = {v[0] / a, v[1] / a, v[2] / a};
    const float b = fmaxf(sqrtf(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]), 1e-6f);
    float* o = XR + (size_t)i * 6;
    for (int j = 0; j < 3; ++j) {
        o[j] = xc[3 * (size_t)i + j];
        o[3 + j] = n1[j] / b;
        nrm_out[3 * (size_t)i + j] = n1[j] / b;
    }
    if (Z8) sdf_out[i] = Z8[(size_t)i * ld];   // Z8 == NULL (with grad given): the caller already holds the sdf column
}

// adjoints: dXA [n][6] (from the colour net), dsdf [n] (from compositing), dnrm_extra [n][3] (direct normal losses, may
// be null) -> column 0 of dZ8 [4P][257] (value rows: d sdf, tangent rows: d grad); the caller zero-fills dZ8 first and the
// colour net's backward accumulates the feature adjoints into columns 1.. of the value rows
__global__ void k_shade_in_bwd(const float* __restrict__ Z8, int P, int n_pts, const float* __restrict__ jinv,
                               const float* __restrict__ dXR, const float* __restrict__ dsdf,
                               const float* __restrict__ dnrm_extra, float* __restrict__ dZ8,
                               float* __restrict__ djinv, const float* __restrict__ grad, float* __restrict__ dgrad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pts) return;
    const int ld = 257;
    const float g[3] = {grad ? grad[3 * (size_t)i] : Z8[(size_t)(P + i) * ld],
                        grad ? grad[3 * (size_t)i + 1] : Z8[(size_t)(2 * P + i) * ld],
                        grad ? grad[3 * (size_t)i + 2] : Z8[(size_t)(3 * P + i) * ld]};
    const float* J = jinv + 9 * (size_t)i;
    float v[3];
    for (int j = 0; j < 3; ++j) v[j] = g[0] * J[j] + g[1] * J[3 + j] + g[2] * J[6 + j];
    const float na = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float a = fmaxf(na, 1e-12f);
    const float n1[3] = {v[0] / a, v[1] / a, v[2] / a};
    const float nb = sqrtf(n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
    const float b = fmaxf(nb, 1e-6f);
    const float n[3] = {n1[0] / b, n1[1] / b, n1[2] / b};
    float dn[3];
    for (int j = 0; j < 3; ++j) dn[j] = dXR[(size_t)i * 6 + 3 + j] + (dnrm_extra ? dnrm_extra[3 * (size_t)i + j] : 0.0f);
    // n = n1 / max(|n1|, eps2)
    float dn1[3];
    {
        const float dot = dn[0] * n[0] + dn[1] * n[1] + dn[2] * n[2];
        for (int j = 0; j < 3; ++j) dn1[j] = nb > 1e-6f ? (dn[j] - n[j] * dot) / b : dn[j] / b;
    }
    // n1 = v / max(|v|, eps1)
    float dv[3];
    {
        const float dot = dn1[0] * n1[0] + dn1[1] * n1[1] + dn1[2] * n1[2];
        for (int j = 0; j < 3; ++j) dv[j] = na > 1e-12f ? (dn1[j] - n1[j] * dot) / a : dn1[j] / a;
    }
    for (int k = 0; k < 3; ++k) {
        const float dg = dv[0] * J[3 * k] + dv[1] * J[3 * k + 1] + dv[2] * J[3 * k + 2];
        if (dgrad) dgrad[3 * (size_t)i + k] = dg;
        else dZ8[(size_t)((k + 1) * P + i) * ld] = dg;
    }
    if (dZ8) dZ8[(size_t)i * ld] = dsdf[i];   // dZ8 == NULL (with dgrad given): the caller passes d sdf on as a vector
    if (djinv)   // v_j = sum_k g_k Jinv[k][j]
        for (int k = 0; k < 3; ++k)
            for (int j = 0; j < 3; ++j) djinv[9 * (size_t)i + 3 * k + j] = g[k] * dv[j];
}

// eikonal points: grad_theta [E][3] = d sdf / d x (raw); rows offset e0 inside the batch of P points
__global__ void k_eik_fwd(const float* __restrict__ Z8, int P, int e0, int E, float* __restrict__ grad_theta,
                          const float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    for (int k = 0; k < 3; ++k)
        grad_theta[3 * (size_t)i + k] = grad ? grad[3 * (size_t)(e0 + i) + k] : Z8[(size_t)((k + 1) * P + e0 + i) * 257];
}
__global__ void k_eik_bwd(int P, int e0, int E, const float* __restrict__ dgrad_theta, float* __restrict__ dZ8,
                          float* __restrict__ dgrad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    for (int k = 1; k <= 3; ++k) {
        if (dgrad) dgrad[3 * (size_t)(e0 + i) + k - 1] = dgrad_theta[3 * (size_t)i + k - 1];
        else dZ8[(size_t)(k * P + e0 + i) * 257] = dgrad_theta[3 * (size_t)i + k - 1];
    }
}

__global__ void k_sigmoid_fwd(const float* __restrict__ Z, long long n, float* __restrict__ Y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Y[i] = 1.0f / (1.0f + expf(-Z[i]));
}
__global__ void k_sigmoid_bwd(const float* __restrict__ Y, const float* __restrict__ dY, long long n, float* __restrict__ dZ) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dZ[i] = dY[i] * Y[i] * (1.0f - Y[i]);
}

// ---- weight norm: W = g v / |v|_row ; also the transposed copy used by the backward-data GEMMs
__global__ __launch_bounds__(64) void k_wn_fwd(const float* __restrict__ v, const float* __restrict__ g, int out_dim,
                                               int in_dim, float* __restrict__ W, float* __restrict__ WT) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f;
    for (int c = lane; c < in_dim; c += 64) ss += v[(size_t)r * in_dim + c] * v[(size_t)r * in_dim + c];
    ss = mp::wsum(ss);
    const float s = g ? g[r] / sqrtf(ss) : 1.0f;
    for (int c = lane; c < in_dim; c += 64) {
        const float w = v[(size_t)r * in_dim + c] * s;
        W[(size_t)r * in_dim + c] = w;
        if (WT) WT[(size_t)c * out_dim + r] = w;
    }
}
// dW -> dv, dg:  dg = dW . v / |v| ; dv = (g/|v|) (dW - (dW . v^) v^)
__global__ __launch_bounds__(64) void k_wn_bwd(const float* __restrict__ v, const float* __restrict__ g, int out_dim,
                                               int in_dim, const float* __restrict__ dW, float* __restrict__ dv,
                                               float* __restrict__ dg) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < in_dim; c += 64) {
        const float x = v[(size_t)r * in_dim + c];
        ss += x * x;
        dot += dW[(size_t)r * in_dim + c] * x;
    }
    ss = mp::wsum(ss);
    dot = mp::wsum(dot);
    const float nrm = sqrtf(ss);
    if (!g) {
        for (int c = lane; c < in_dim; c += 64) dv[(size_t)r * in_dim + c] = dW[(size_t)r * in_dim + c];
        return;
    }
    const float gg = g[r];
    for (int c = lane; c < in_dim; c += 64)
        dv[(size_t)r * in_dim + c] = (gg / nrm) * (dW[(size_t)r * in_dim + c] - dot / ss * v[(size_t)r * in_dim + c]);
    if (lane == 0) dg[r] = dot / nrm;
}

// The same two kernels for MANY layers in one launch (a training iteration resolves ~40 weight-normed layers: 2 launches instead
// of ~80): block = one row of the concatenated row range, its layer found by binary search over the descriptors' first rows.
__device__ __forceinline__ const MpWnDesc& wn_find(const MpWnDesc* __restrict__ d, int n, int row, int& r) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].row0 <= row) lo = mid; else hi = mid - 1;
    }
    r = row - d[lo].row0;
    return d[lo];
}
__global__ __launch_bounds__(64) void k_wn_fwd_multi(const MpWnDesc* __restrict__ descs, int n) {
    int r;
    const MpWnDesc& D = wn_find(descs, n, blockIdx.x, r);
    const int lane = threadIdx.x, in_dim = D.in_dim, out_dim = D.out_dim;
    const float* v = D.v + (size_t)r * in_dim;
    float ss = 0.f;
    if (D.g)
        for (int c = lane; c < in_dim; c += 64) ss += v[c] * v[c];
    ss = mp::wsum(ss);
    const float s = D.g ? D.g[r] / sqrtf(ss) : 1.0f;
    for (int c = lane; c < in_dim; c += 64) {
        const float w = v[c] * s;
        D.W[(size_t)r * in_dim + c] = w;
        if (D.WT) D.WT[(size_t)c * out_dim + r] = w;
    }
}
__global__ __launch_bounds__(64) void k_wn_bwd_multi(const MpWnDesc* __restrict__ descs, int n, const float* __restrict__ acc_base,
                                                      float* __restrict__ grad_base) {
    int r;
    const MpWnDesc& D = wn_find(descs, n, blockIdx.x, r);
    const int lane = threadIdx.x, in_dim = D.in_dim;
    const float* v = D.v + (size_t)r * in_dim;
    const float* dW = acc_base + D.dW_off + (size_t)r * in_dim;
    float* dv = grad_base + D.dv_off + (size_t)r * in_dim;
    if (!D.g) {
        for (int c = lane; c < in_dim; c += 64) dv[c] = dW[c];
        return;
    }
    float ss = 0.f, dot = 0.f;
    for (int c = lane; c < in_dim; c += 64) {
        const float x = v[c];
        ss += x * x;
        dot += dW[c] * x;
    }
    ss = mp::wsum(ss);
    dot = mp::wsum(dot);
    const float nrm = sqrtf(ss), gg = D.g[r];
    for (int c = lane; c < in_dim; c += 64) dv[c] = (gg / nrm) * (dW[c] - dot / ss * v[c]);
    if (lane == 0) grad_base[D.dg_off + r] = dot / nrm;
}

// hoisted conditioning: b2[r] = b[r] + sum_c W[r][c0+c] vec[c]   /  dW[r][c0+c] += db2[r] vec[c]
__global__ __launch_bounds__(64) void k_hoist_fwd(const float* __restrict__ W, int in_dim, const float* __restrict__ b, int c0,
                                                  int n, const float* __restrict__ vec, float* __restrict__ b2) {
    const int r = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += W[(size_t)r * in_dim + c0 + c] * vec[c];
    s = mp::wsum(s);
    if (lane == 0) b2[r] = b[r] + s;
}
__global__ void k_hoist_bwd(const float* __restrict__ db2, int out_dim, int in_dim, int c0, int n,
                            const float* __restrict__ vec, float* __restrict__ dW) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= out_dim * n) return;
    const int r = idx / n, c = idx % n;
    dW[(size_t)r * in_dim + c0 + c] += db2[r] * vec[c];
}

// column sums of the first `rows` rows: db[c] = sum_r dZ[r][c]
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ dZ, int ld, int rows, int C, float* __restrict__ db) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) s += dZ[(size_t)r * ld + c];
    s = mp::wsum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[c] = sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- compositing backward.  Like the forward (composite.hip) ONE WAVE owns one ray and nothing is merged: with fe the free
// energy sigma dt of a sample, E its exclusive sum in the merged order, T = exp(-E), w = (1 - exp(-fe)) T and
// dw = dC . rgb + dA + dA_p, the adjoint of fe_j is
//     dw_j T_j exp(-fe_j)  -  sum_{i behind j} dw_i w_i  -  [j is not the very last sample] dT_bg T_bg
// "In front of / behind (p, j)" in another person q's sorted list is a rank (binary search on t_end; ties: lower person
// first, the order of the reference's stable sorts), so E and the suffix sum are per-person prefix sums looked up at those
// ranks.  (The first version walked the merged list with one THREAD per ray: 512 threads on the whole device, ~1 ms.)
constexpr int MAX_P = 8;
__global__ __launch_bounds__(256) void k_composite_bwd(
    int n_rays, int P, int n_z, const int* const* __restrict__ inv_index, const float* const* __restrict__ z,
    const float* const* __restrict__ sdf, const float* const* __restrict__ rgb, const float* __restrict__ beta_p,
    const float* __restrict__ bg_rgb, const float* __restrict__ d_rgb_values, const float* __restrict__ d_acc,
    const float* __restrict__ d_acc_person, float* const* __restrict__ d_sdf, float* const* __restrict__ d_rgb,
    float* __restrict__ d_bg_rgb, float* __restrict__ d_beta) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int r = blockIdx.x * wpb + wave;
    const int S = n_z - 1;
    // per wave, per person: te [S], fe [S], pf [S+1] (prefix of fe), w [S], gp [S+1] (prefix of dw w)
    const int per_person = 5 * S + 2;
    float* base = smem + (size_t)wave * P * per_person;
    const float beta = *beta_p;
    const bool live = r < n_rays;
    int k[MAX_P];
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) k[p] = (live && p < P) ? inv_index[p][r] : -1;
    float dC[3] = {0.f, 0.f, 0.f}, dA = 0.f;
    if (live) {
        for (int a = 0; a < 3; ++a) dC[a] = d_rgb_values[3 * r + a];
        dA = d_acc ? d_acc[r] : 0.0f;
    }
    // ---- phase 1: free energies and their per-person prefix sums
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
    // rank of t_end `te` of person p's sample among person q's samples (q != p)
    auto rank_in = [&](int q, int p, float te) {
        const float* te_q = base + q * per_person;
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float t = te_q[mid];
            const bool before = q < p ? t <= te : t < te;
            if (before) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    // ---- phase 2: weights, dw w and its per-person prefix sums
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) {
        if (p < P && k[p] >= 0) {
            float* te_l = base + p * per_person;
            float* fe_l = te_l + S;
            float* pf_l = fe_l + S;
            float* w_l = pf_l + S + 1;
            float* gp_l = w_l + S;
            const float dAp = d_acc_person ? d_acc_person[(size_t)r * P + p] : 0.0f;
            float carry = 0.f;
            if (lane == 0) gp_l[0] = 0.f;
            for (int i0 = 0; i0 < S; i0 += 64) {
                const int i = i0 + lane;
                float g = 0.f;
                if (i < S) {
                    const float te = te_l[i];
                    float E = pf_l[i];
#pragma unroll
                    for (int q = 0; q < MAX_P; ++q)
                        if (q < P && q != p && k[q] >= 0) E += (base + q * per_person + 2 * S)[rank_in(q, p, te)];
                    const float T = expf(-E);
                    const float w = (1.0f - expf(-fe_l[i])) * T;
                    const size_t qi = (size_t)k[p] * S + i;
                    const float dw = dC[0] * rgb[p][3 * qi] + dC[1] * rgb[p][3 * qi + 1] + dC[2] * rgb[p][3 * qi + 2] + dA + dAp;
                    w_l[i] = w;
                    g = dw * w;
                    d_rgb[p][3 * qi] = w * dC[0]; d_rgb[p][3 * qi + 1] = w * dC[1]; d_rgb[p][3 * qi + 2] = w * dC[2];
                }
                float tot;
                const float ex = mp::wave_excl_scan(g, tot);
                if (i < S) gp_l[i + 1] = carry + ex + g;
                carry += tot;
            }
        }
    }
    __syncthreads();
    // the very last sample of the merged order, and T_bg = its exclusive transmittance (multiply.py:457-463)
    int p_last = -1;
    float tm = -FLT_MAX;
    for (int p = 0; p < P; ++p)
        if (k[p] >= 0 && (base + p * per_person)[S - 1] >= tm) { tm = (base + p * per_person)[S - 1]; p_last = p; }
    float E_front = 0.f;
    for (int p = 0; p < P; ++p)
        if (k[p] >= 0) E_front += (base + p * per_person + 2 * S)[p == p_last ? S - 1 : S];
    const float Tbg = p_last >= 0 ? expf(-E_front) : 1.0f;
    float dTbg = 0.f;
    for (int a = 0; a < 3; ++a) dTbg += dC[a] * (bg_rgb && live ? bg_rgb[3 * r + a] : 1.0f);
    if (live && lane < 3 && d_bg_rgb) d_bg_rgb[3 * r + lane] = dC[lane] * Tbg;
    // ---- phase 3: adjoints of the free energies -> sdf and beta
    float dbeta_local = 0.0f;
#pragma unroll
    for (int p = 0; p < MAX_P; ++p) {
        if (p < P && k[p] >= 0) {
            const float* te_l = base + p * per_person;
            const float* fe_l = te_l + S;
            const float* pf_l = fe_l + S;
            const float* w_l = pf_l + S + 1;
            const float* gp_l = w_l + S;
            const float* zr = z[p] + (size_t)k[p] * n_z;
            for (int i = lane; i < S; i += 64) {
                const float te = te_l[i], fe = fe_l[i];
                float E = pf_l[i];
                float suffix = gp_l[S] - gp_l[i + 1];
#pragma unroll
                for (int q = 0; q < MAX_P; ++q)
                    if (q < P && q != p && k[q] >= 0) {
                        const int lo = rank_in(q, p, te);
                        const float* pq = base + q * per_person + 2 * S;
                        E += pq[lo];
                        suffix += (pq + 2 * S + 1)[S] - (pq + 2 * S + 1)[lo];
                    }
                const float T = expf(-E), ex = expf(-fe);
                const size_t qi = (size_t)k[p] * S + i;
                const float dw = dC[0] * rgb[p][3 * qi] + dC[1] * rgb[p][3 * qi + 1] + dC[2] * rgb[p][3 * qi + 2] + dA +
                                 (d_acc_person ? d_acc_person[(size_t)r * P + p] : 0.0f);
                float dfe = dw * T * ex - suffix;
                if (!(p == p_last && i == S - 1)) dfe -= dTbg * Tbg;   // T_bg depends on every sample but the last
                const float s = sdf[p][qi];
                const float dsig = dfe * (te - zr[i]);
                // d sigma / d sdf and d sigma / d beta (density.py:20-29)
                const float eab = expf(-fabsf(s) / beta);
                const float dsdf = -(0.5f / (beta * beta)) * eab;
                float dbe;
                if (s > 0.f) dbe = (0.5f / (beta * beta)) * eab * (s / beta - 1.0f);
                else if (s < 0.f) dbe = -1.0f / (beta * beta) + (0.5f / (beta * beta)) * eab * (1.0f + s / beta);
                else dbe = -0.5f / (beta * beta);
                d_sdf[p][qi] = dsig * dsdf;
                dbeta_local += dsig * dbe;
            }
        }
    }
    dbeta_local = mp::wsum(dbeta_local);
    if (lane == 0 && dbeta_local != 0.0f) atomicAdd(d_beta, dbeta_local);
}

// ---- background compositing (multiply.py:682-696) forward with stash-free backward; one thread per ray
__global__ void k_bg_comp_fwd(const float* __restrict__ sdf, const float* __restrict__ rgb, const float* __restrict__ zbg,
                              int R, int NBG, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float csum = 0.f, acc[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < NBG; ++i) {
        const float dist = i + 1 < NBG ? zbg[(size_t)r * NBG + i] - zbg[(size_t)r * NBG + i + 1] : 1e10f;
        const float fe = dist * fabsf(sdf[(size_t)r * NBG + i]);
        const float w = (1.0f - expf(-fe)) * expf(-csum);
        for (int a = 0; a < 3; ++a) acc[a] += w * rgb[3 * ((size_t)r * NBG + i) + a];
        csum += fe;
    }
    for (int a = 0; a < 3; ++a) out[3 * r + a] = acc[a];
}
__global__ void k_bg_comp_bwd(const float* __restrict__ sdf, const float* __restrict__ rgb, const float* __restrict__ zbg,
                              int R, int NBG, const float* __restrict__ dout, float* __restrict__ dsdf,
                              float* __restrict__ drgb) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    // E = free energy in front of sample i; starts at the last sample (whose own 1e10-long interval must never enter the
    // running sum: it would swallow every other term in fp32)
    float E = 0.f;
    for (int i = 0; i + 1 < NBG; ++i)
        E += (zbg[(size_t)r * NBG + i] - zbg[(size_t)r * NBG + i + 1]) * fabsf(sdf[(size_t)r * NBG + i]);
    const float dC[3] = {dout[3 * r], dout[3 * r + 1], dout[3 * r + 2]};
    float suffix = 0.f;
    for (int i = NBG - 1; i >= 0; --i) {
        const size_t q = (size_t)r * NBG + i;
        const float dist = i + 1 < NBG ? zbg[q] - zbg[q + 1] : 1e10f;
        const float s = sdf[q];
        const float fe = dist * fabsf(s);
        if (i + 1 < NBG) E -= fe;
        const float T = expf(-E), ex = expf(-fe), w = (1.0f - ex) * T;
        const float dw = dC[0] * rgb[3 * q] + dC[1] * rgb[3 * q + 1] + dC[2] * rgb[3 * q + 2];
        const float dfe = dw * T * ex - suffix;
        dsdf[q] = dfe * dist * (s > 0.f ? 1.0f : (s < 0.f ? -1.0f : 0.0f));
        for (int a = 0; a < 3; ++a) drgb[3 * q + a] = w * dC[a];
        suffix += dw * w;
    }
}

// NeRF++ background points (multiply.py:698-726) as plain arrays: pts [R*NBG][4], per-ray depths zbg [R][NBG]
__global__ void k_bg_points(const float* __restrict__ dirs, const float* __restrict__ cam, const float* __restrict__ zbg,
                            int R, int NBG, float radius, float* __restrict__ pts) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long long)R * NBG) return;
    const int ray = (int)(q / NBG);
    const float d[3] = {dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]};
    const float ox = cam[0], oy = cam[1], oz = cam[2];
    const float depth = zbg[q];
    const float o_dot_d = d[0] * ox + d[1] * oy + d[2] * oz;
    const float under = o_dot_d * o_dot_d - ((ox * ox + oy * oy + oz * oz) - radius * radius);
    const float d_sphere = sqrtf(under) - o_dot_d;
    const float ps[3] = {ox + d_sphere * d[0], oy + d_sphere * d[1], oz + d_sphere * d[2]};
    const float pm[3] = {ox - o_dot_d * d[0], oy - o_dot_d * d[1], oz - o_dot_d * d[2]};
    const float pm_n = sqrtf(pm[0] * pm[0] + pm[1] * pm[1] + pm[2] * pm[2]);
    float ax[3] = {oy * ps[2] - oz * ps[1], oz * ps[0] - ox * ps[2], ox * ps[1] - oy * ps[0]};
    const float an = sqrtf(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    ax[0] /= an; ax[1] /= an; ax[2] /= an;
    const float phi = asinf(pm_n / radius), theta = asinf(pm_n * depth);
    float sa, ca;
    sincosf(phi - theta, &sa, &ca);
    const float cr[3] = {ax[1] * ps[2] - ax[2] * ps[1], ax[2] * ps[0] - ax[0] * ps[2], ax[0] * ps[1] - ax[1] * ps[0]};
    const float adp = ax[0] * ps[0] + ax[1] * ps[1] + ax[2] * ps[2];
    float pn[3];
    for (int a = 0; a < 3; ++a) pn[a] = ps[a] * ca + cr[a] * sa + ax[a] * adp * (1.0f - ca);
    const float pnn = sqrtf(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
    for (int a = 0; a < 3; ++a) pts[4 * q + a] = pn[a] / pnn;
    pts[4 * q + 3] = depth;
}

__global__ void k_copy_cols(const float* __restrict__ src, int lds, int c0s, float* __restrict__ dst, int ldd, int c0d,
                            long long rows, int C, float scale, int accumulate) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const size_t r = idx / C;
    const int c = (int)(idx % C);
    const float v = src[r * lds + c0s + c] * scale;
    float* d = dst + r * ldd + c0d + c;
    *d = accumulate ? *d + v : v;
}


// ---- adjoint of the Fourier features w.r.t. the point: dIN [(fwd?4:1)*P][ld] -> dx [P][D] (+=)
//   value row : x_a, sin(f x_a), cos(f x_a);  tangent row a : 1, f cos(f x_a), -f sin(f x_a)  (only dimension a's columns)
template <int D>
__global__ void k_pe_bwd(const float* __restrict__ x, int P, int L, int fwd, const float* __restrict__ dIN, int ld,
                         float* __restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float* dv = dIN + (size_t)i * ld;
    for (int a = 0; a < D; ++a) {
        const float xa = x[(size_t)i * D + a];
        const float* dt = fwd ? dIN + (size_t)((a + 1) * P + i) * ld : nullptr;
        float acc = dv[a];
        for (int k = 0; k < L; ++k) {
            const float f = (float)(1 << k);
            float sn, cs;
            sincosf(xa * f, &sn, &cs);
            const int cs_ = D + 2 * D * k + a, cc_ = cs_ + D;
            acc += f * (cs * dv[cs_] - sn * dv[cc_]);
            if (fwd) acc -= f * f * (sn * dt[cs_] + cs * dt[cc_]);
        }
        dx[(size_t)i * D + a] += acc;
    }
}

// ---- adjoint of the canonical warp w.r.t. the bone transforms (deformer.py:19-50, 72-88; multiply.py:625-641)
//   x_c = R^-1 (x - t),  [R t] = sum_j w_j tfs_j  (w = skinning weights of the nearest POSED vertex, detached)
//   Jinv = Rc^-1,        Rc    = sum_j wc_j tfs_j[:3,:3]  (wc = weights of the nearest CANONICAL vertex)
//   d R = -R^-T dx_c x_c^T ; d t = -R^-T dx_c ; d Rc = -Jinv^T dJinv Jinv^T ; d tfs_j += w_j [dR dt] + wc_j [dRc 0]
// Accumulated per workgroup in LDS (24 x 12 floats), flushed with one atomic per entry.
__global__ __launch_bounds__(256) void k_warp_bwd(const float* __restrict__ xc, const float* __restrict__ dxc,
                                                  const float* __restrict__ jinv, const float* __restrict__ djinv,
                                                  const int* __restrict__ nn_posed, const int* __restrict__ nn_cano, int n,
                                                  const float* __restrict__ skin_w, const float* __restrict__ tfs,
                                                  float* __restrict__ dtfs) {
    constexpr int NJ = 24;
    __shared__ float acc[NJ * 12];
    __shared__ float tl[NJ * 16];
    for (int i = threadIdx.x; i < NJ * 12; i += blockDim.x) acc[i] = 0.f;
    for (int i = threadIdx.x; i < NJ * 16; i += blockDim.x) tl[i] = tfs[i];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float* w = skin_w + (size_t)nn_posed[i] * NJ;
        float R[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < NJ; ++j) {
            const float wj = w[j];
            if (wj != 0.f)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) R[3 * a + b] += wj * tl[16 * j + 4 * a + b];
        }
        // inverse of R
        const float c0 = R[4] * R[8] - R[5] * R[7], c1 = R[5] * R[6] - R[3] * R[8], c2 = R[3] * R[7] - R[4] * R[6];
        const float r = 1.0f / (R[0] * c0 + R[1] * c1 + R[2] * c2);
        const float I[9] = {c0 * r, (R[2] * R[7] - R[1] * R[8]) * r, (R[1] * R[5] - R[2] * R[4]) * r,
                            c1 * r, (R[0] * R[8] - R[2] * R[6]) * r, (R[2] * R[3] - R[0] * R[5]) * r,
                            c2 * r, (R[1] * R[6] - R[0] * R[7]) * r, (R[0] * R[4] - R[1] * R[3]) * r};
        const float d[3] = {dxc[3 * (size_t)i], dxc[3 * (size_t)i + 1], dxc[3 * (size_t)i + 2]};
        const float q[3] = {xc[3 * (size_t)i], xc[3 * (size_t)i + 1], xc[3 * (size_t)i + 2]};
        float u[3];   // u = R^-T dx_c
        for (int a = 0; a < 3; ++a) u[a] = I[a] * d[0] + I[3 + a] * d[1] + I[6 + a] * d[2];
        float dT[12];
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) dT[4 * a + b] = -u[a] * q[b];
            dT[4 * a + 3] = -u[a];
        }
        for (int j = 0; j < NJ; ++j) {
            const float wj = w[j];
            if (wj != 0.f)
                for (int e = 0; e < 12; ++e) atomicAdd(&acc[12 * j + e], wj * dT[e]);
        }
        if (djinv) {
            const float* M = jinv + 9 * (size_t)i;
            const float* dM = djinv + 9 * (size_t)i;
            float t1[9], dRc[9];   // dRc = -M^T dM M^T
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    float sacc = 0.f;
                    for (int k = 0; k < 3; ++k) sacc += M[3 * k + a] * dM[3 * k + b];
                    t1[3 * a + b] = sacc;
                }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    float sacc = 0.f;
                    for (int k = 0; k < 3; ++k) sacc += t1[3 * a + k] * M[3 * b + k];
                    dRc[3 * a + b] = -sacc;
                }
            const float* wc = skin_w + (size_t)nn_cano[i] * NJ;
            for (int j = 0; j < NJ; ++j) {
                const float wj = wc[j];
                if (wj != 0.f)
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) atomicAdd(&acc[12 * j + 4 * a + b], wj * dRc[3 * a + b]);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NJ * 12; e += blockDim.x)
        if (acc[e] != 0.f) atomicAdd(&dtfs[16 * (e / 12) + (e % 12)], acc[e]);
}

// ---- adjoint of SMPLServer.forward's bone transforms (smpl.py:50-94, lbs.py:276-377) w.r.t. the 86 SMPL parameters
//   [scale, transl(3), thetas(72), betas(10)];  one thread: 24 joints, a few hundred flops each.
//   Only the transforms are differentiated: the posed vertices enter the hot path through a nearest-vertex index.
__global__ void k_smpl_pose_bwd(const int* __restrict__ parents, const float* __restrict__ params,
                                const float* __restrict__ tfs_c_inv, const float* __restrict__ rest_joints,
                                const float* __restrict__ j_shapedirs, const float* __restrict__ dtfs,
                                float* __restrict__ dparams) {
    constexpr int NJ = 24;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float scale = params[0];
    const float* transl = params + 1;
    const float* th = params + 4;
    const float* J = rest_joints;
    float R[NJ][9], G[NJ][12], dG[NJ][12], dJ[NJ][3], dR[NJ][9];
    for (int j = 0; j < NJ; ++j) {
        const float ax = th[3 * j] + 1e-8f, ay = th[3 * j + 1] + 1e-8f, az = th[3 * j + 2] + 1e-8f;
        const float ang = sqrtf(ax * ax + ay * ay + az * az);
        const float n[3] = {th[3 * j] / ang, th[3 * j + 1] / ang, th[3 * j + 2] / ang};
        float s, c;
        sincosf(ang, &s, &c);
        const float K[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                float kk = 0.f;
                for (int k = 0; k < 3; ++k) kk += K[3 * a + k] * K[3 * k + b];
                R[j][3 * a + b] = (a == b ? 1.f : 0.f) + s * K[3 * a + b] + (1.f - c) * kk;
            }
        const int p = parents[j];
        float rel[3];
        for (int a = 0; a < 3; ++a) rel[a] = J[3 * j + a] - (j > 0 ? J[3 * p + a] : 0.f);
        if (j == 0) {
            for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) G[0][4 * a + b] = R[0][3 * a + b]; G[0][4 * a + 3] = rel[a]; }
        } else {
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) {
                    float v = 0.f;
                    for (int k = 0; k < 3; ++k) v += G[p][4 * a + k] * R[j][3 * k + b];
                    G[j][4 * a + b] = v;
                }
                float v = G[p][4 * a + 3];
                for (int k = 0; k < 3; ++k) v += G[p][4 * a + k] * rel[k];
                G[j][4 * a + 3] = v;
            }
        }
        for (int a = 0; a < 3; ++a) dJ[j][a] = 0.f;
    }
    float dscale = 0.f, dtr[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < NJ; ++j) {
        // tfs_j = tf_j C_j  ->  d tf = dtfs C^T  (rows 0..2; C's last row is [0,0,0,1] for the absolute case C = I)
        float dtf[12];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 4; ++b) {
                float v = 0.f;
                if (tfs_c_inv) for (int k = 0; k < 4; ++k) v += dtfs[16 * j + 4 * a + k] * tfs_c_inv[16 * j + 4 * b + k];
                else v = dtfs[16 * j + 4 * a + b];
                dtf[4 * a + b] = v;
            }
        // A = G with translation column  A[a][3] = G[a][3] - sum_k G[a][k] J_j[k];  tf = scale A, tf[a][3] += scale transl[a]
        float dA[12];
        for (int a = 0; a < 3; ++a) {
            float A3 = G[j][4 * a + 3];
            for (int k = 0; k < 3; ++k) A3 -= G[j][4 * a + k] * J[3 * j + k];
            for (int b = 0; b < 3; ++b) dscale += dtf[4 * a + b] * G[j][4 * a + b];
            dscale += dtf[4 * a + 3] * (A3 + transl[a]);
            dtr[a] += scale * dtf[4 * a + 3];
            for (int b = 0; b < 4; ++b) dA[4 * a + b] = scale * dtf[4 * a + b];
        }
        for (int a = 0; a < 3; ++a) {
            for (int k = 0; k < 3; ++k) {
                dG[j][4 * a + k] = dA[4 * a + k] - dA[4 * a + 3] * J[3 * j + k];
                dJ[j][k] -= dA[4 * a + 3] * G[j][4 * a + k];
            }
            dG[j][4 * a + 3] = dA[4 * a + 3];
        }
    }
    for (int j = NJ - 1; j >= 1; --j) {   // G_j = G_p [R_j | rel_j]
        const int p = parents[j];
        float rel[3];
        for (int a = 0; a < 3; ++a) rel[a] = J[3 * j + a] - J[3 * p + a];
        float drel[3] = {0.f, 0.f, 0.f};
        for (int b = 0; b < 3; ++b)
            for (int c2 = 0; c2 < 3; ++c2) {
                float v = 0.f;
                for (int a = 0; a < 3; ++a) v += G[p][4 * a + b] * dG[j][4 * a + c2];
                dR[j][3 * b + c2] = v;
            }
        for (int b = 0; b < 3; ++b)
            for (int a = 0; a < 3; ++a) drel[b] += G[p][4 * a + b] * dG[j][4 * a + 3];
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) {
                float v = dG[j][4 * a + 3] * rel[b];
                for (int c2 = 0; c2 < 3; ++c2) v += dG[j][4 * a + c2] * R[j][3 * b + c2];
                dG[p][4 * a + b] += v;
            }
            dG[p][4 * a + 3] += dG[j][4 * a + 3];
        }
        for (int a = 0; a < 3; ++a) { dJ[j][a] += drel[a]; dJ[p][a] -= drel[a]; }
    }
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) dR[0][3 * a + b] = dG[0][4 * a + b];
        dJ[0][a] += dG[0][4 * a + 3];
    }
    for (int i = 0; i < 86; ++i) dparams[i] = 0.f;
    dparams[0] = dscale;
    for (int a = 0; a < 3; ++a) dparams[1 + a] = dtr[a];
    for (int j = 0; j < NJ; ++j) {   // Rodrigues with angle = |theta + 1e-8|, axis = theta / angle (lbs.py:290-296)
        const float t3[3] = {th[3 * j], th[3 * j + 1], th[3 * j + 2]};
        const float e3[3] = {t3[0] + 1e-8f, t3[1] + 1e-8f, t3[2] + 1e-8f};
        const float ang = sqrtf(e3[0] * e3[0] + e3[1] * e3[1] + e3[2] * e3[2]);
        const float n[3] = {t3[0] / ang, t3[1] / ang, t3[2] / ang};
        float s, c;
        sincosf(ang, &s, &c);
        const float K[9] = {0.f, -n[2], n[1], n[2], 0.f, -n[0], -n[1], n[0], 0.f};
        float KK[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                float kk = 0.f;
                for (int k = 0; k < 3; ++k) kk += K[3 * a + k] * K[3 * k + b];
                KK[3 * a + b] = kk;
            }
        float dang = 0.f;
        for (int e = 0; e < 9; ++e) dang += dR[j][e] * (c * K[e] + s * KK[e]);
        // dK (adjoint of K): from s K and (1-c) K K  ->  dK = s dR + (1-c) (dR K^T + K^T dR)
        float dK[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                float v = s * dR[j][3 * a + b];
                for (int k = 0; k < 3; ++k) v += (1.f - c) * (dR[j][3 * a + k] * K[3 * b + k] + K[3 * k + a] * dR[j][3 * k + b]);
                dK[3 * a + b] = v;
            }
        const float dn[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
        float ndot = 0.f;
        for (int k = 0; k < 3; ++k) ndot += dn[k] * t3[k];
        for (int i = 0; i < 3; ++i)
            dparams[4 + 3 * j + i] = dang * e3[i] / ang + dn[i] / ang - ndot * e3[i] / (ang * ang * ang);
    }
    if (j_shapedirs)
        for (int l = 0; l < 10; ++l) {
            float v = 0.f;
            for (int j = 0; j < NJ; ++j)
                for (int k = 0; k < 3; ++k) v += dJ[j][k] * j_shapedirs[(3 * j + k) * 10 + l];
            dparams[76 + l] = v;
        }
}


// ---- reverse-over-reverse SDF net (multiply_amd/train.py ImplicitTrainRev): the spatial gradient of the sdf comes from a
// reverse sweep V_l = sigma'(Z_l) (.) U_l, U_{l-1} = V_l W_l, and the training backward is the adjoint of BOTH sweeps:
// 6 GEMMs per layer over P rows instead of 3 GEMMs over the 4P rows of the forward-mode formulation.
//   sigmul    : V = sigma'(Z) (.) U * scale            (U == NULL: U = wrow broadcast over the rows)
//   rev_adj   : dU = sigma'(Z) (.) dV ;  dS = U (.) dV   (adjoint of sigmul w.r.t. U and w.r.t. sigma')
//   dz        : dZ = sigma'(Z) (.) dX * scale + sigma''(Z) (.) dS
__global__ void k_sigmul(const float* __restrict__ Z, int ldz, long long rows, int C, const float* __restrict__ U, int ldu,
                         const float* __restrict__ wrow, float scale, float* __restrict__ V, int ldv) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const size_t r = idx / C;
    const int c = (int)(idx % C);
    float h, d1, d2;
    softplus_d012(Z[r * ldz + c], h, d1, d2);
    V[r * ldv + c] = d1 * (U ? U[r * ldu + c] : wrow[c]) * scale;
}
__global__ void k_rev_adj(const float* __restrict__ Z, int ldz, long long rows, int C, const float* __restrict__ U, int ldu,
                          const float* __restrict__ wrow, float uscale, const float* __restrict__ dV, int lddv,
                          float* __restrict__ dU, int lddu, float* __restrict__ dS, int ldds) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const size_t r = idx / C;
    const int c = (int)(idx % C);
    float h, d1, d2;
    softplus_d012(Z[r * ldz + c], h, d1, d2);
    const float dv = dV[r * lddv + c];
    dU[r * lddu + c] = d1 * dv;
    dS[r * ldds + c] = (U ? U[r * ldu + c] : wrow[c]) * uscale * dv;
}
__global__ void k_dz(const float* __restrict__ Z, int ldz, long long rows, int C, const float* __restrict__ dX, int lddx,
                     float scale, const float* __restrict__ dS, int ldds, float* __restrict__ dZ, int lddz) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const size_t r = idx / C;
    const int c = (int)(idx % C);
    float h, d1, d2;
    softplus_d012(Z[r * ldz + c], h, d1, d2);
    dZ[r * lddz + c] = d1 * dX[r * lddx + c] * scale + d2 * dS[r * ldds + c];
}
// ---- the same five element-wise passes, FOUR columns per thread (float4 loads / stores, 32-bit index arithmetic): taken when
// every pointer is 16-byte aligned and C, the leading dimensions and the column offset are multiples of 4 -- always, for the
// 256-wide hidden layers.  The one-element-per-thread forms above (a 64-bit division per element, 4-byte accesses) ran at ~2
// of the ~4.5 TB/s these passes can stream at.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool ew4_index(unsigned rows, unsigned C4, unsigned& r, unsigned& c) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C4) return false;
    r = i / C4;
    c = (i - r * C4) * 4;
    return true;
}
__global__ void k_softplus_fwd4(const float* __restrict__ Z, int ldz, unsigned rows, unsigned C4, float scale, float* __restrict__ H,
                                int ldh, int col0) {
    unsigned r, c;
    if (!ew4_index(rows, C4, r, c)) return;
    const f4v z = *(const f4v*)(Z + (size_t)r * ldz + c);
    f4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float h, d1, d2; softplus_d012(z[e], h, d1, d2); o[e] = h * scale; }
    *(f4v*)(H + (size_t)r * ldh + col0 + c) = o;
}
__global__ void k_relu_bwd4(const float* __restrict__ H, int ldh, unsigned rows, unsigned C4, const float* __restrict__ dH, int lddh,
                            float* __restrict__ dZ, int lddz) {
    unsigned r, c;
    if (!ew4_index(rows, C4, r, c)) return;
    const f4v h = *(const f4v*)(H + (size_t)r * ldh + c), d = *(const f4v*)(dH + (size_t)r * lddh + c);
    f4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = h[e] > 0.0f ? d[e] : 0.0f;
    *(f4v*)(dZ + (size_t)r * lddz + c) = o;
}
__global__ void k_sigmul4(const float* __restrict__ Z, int ldz, unsigned rows, unsigned C4, const float* __restrict__ U, int ldu,
                          const float* __restrict__ wrow, float scale, float* __restrict__ V, int ldv) {
    unsigned r, c;
    if (!ew4_index(rows, C4, r, c)) return;
    const f4v z = *(const f4v*)(Z + (size_t)r * ldz + c);
    const f4v u = U ? *(const f4v*)(U + (size_t)r * ldu + c) : *(const f4v*)(wrow + c);
    f4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float h, d1, d2; softplus_d012(z[e], h, d1, d2); o[e] = d1 * u[e] * scale; }
    *(f4v*)(V + (size_t)r * ldv + c) = o;
}
__global__ void k_rev_adj4(const float* __restrict__ Z, int ldz, unsigned rows, unsigned C4, const float* __restrict__ U, int ldu,
                           const float* __restrict__ wrow, float uscale, const float* __restrict__ dV, int lddv,
                           float* __restrict__ dU, int lddu, float* __restrict__ dS, int ldds) {
    unsigned r, c;
    if (!ew4_index(rows, C4, r, c)) return;
    const f4v z = *(const f4v*)(Z + (size_t)r * ldz + c), dv = *(const f4v*)(dV + (size_t)r * lddv + c);
    const f4v u = U ? *(const f4v*)(U + (size_t)r * ldu + c) : *(const f4v*)(wrow + c);
    f4v ou, os;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float h, d1, d2;
        softplus_d012(z[e], h, d1, d2);
        ou[e] = d1 * dv[e];
        os[e] = u[e] * uscale * dv[e];
    }
    *(f4v*)(dU + (size_t)r * lddu + c) = ou;
    *(f4v*)(dS + (size_t)r * ldds + c) = os;
}
__global__ void k_dz4(const float* __restrict__ Z, int ldz, unsigned rows, unsigned C4, const float* __restrict__ dX, int lddx,
                      float scale, const float* __restrict__ dS, int ldds, float* __restrict__ dZ, int lddz) {
    unsigned r, c;
    if (!ew4_index(rows, C4, r, c)) return;
    const f4v z = *(const f4v*)(Z + (size_t)r * ldz + c), dx = *(const f4v*)(dX + (size_t)r * lddx + c),
              ds = *(const f4v*)(dS + (size_t)r * ldds + c);
    f4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { float h, d1, d2; softplus_d012(z[e], h, d1, d2); o[e] = d1 * dx[e] * scale + d2 * ds[e]; }
    *(f4v*)(dZ + (size_t)r * lddz + c) = o;
}
inline bool al16(const void* p) { return ((size_t)p & 15) == 0; }
inline bool mul4(long long a) { return (a & 3) == 0; }
inline bool fits32(long long rows, int C) { return rows > 0 && rows * (C / 4) < 0x7fffffffLL; }
inline dim3 grid4(long long rows, int C) { return grid1(rows * (C / 4)); }

// Fourier-feature Jacobian (3-D points, L octaves; embedders.py layout): grad[a] = sum_f G[f] dPE_f/dx_a ;
// adjoint: dG[f] = dgrad[a(f)] dPE_f/dx_a ; dx[a] += sum_f G[f] dgrad[a] d2PE_f/dx_a^2   (optional)
__global__ void k_pe_grad_fwd(const float* __restrict__ x, int P, int L, const float* __restrict__ G, int ldg,
                              float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float* g = G + (size_t)i * ldg;
    for (int a = 0; a < 3; ++a) {
        const float xa = x[3 * (size_t)i + a];
        float acc = g[a];
        for (int k = 0; k < L; ++k) {
            const float f = (float)(1 << k);
            float sn, cs;
            sincosf(xa * f, &sn, &cs);
            acc += f * (cs * g[3 + 6 * k + a] - sn * g[3 + 6 * k + 3 + a]);
        }
        grad[3 * (size_t)i + a] = acc;
    }
}
__global__ void k_pe_grad_bwd(const float* __restrict__ x, int P, int L, const float* __restrict__ dgrad,
                              const float* __restrict__ G, int ldg, float* __restrict__ dG, int lddg, float* __restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    float* dg = dG + (size_t)i * lddg;
    for (int a = 0; a < 3; ++a) {
        const float xa = x[3 * (size_t)i + a], da = dgrad[3 * (size_t)i + a];
        dg[a] = da;
        float acc = 0.f;
        for (int k = 0; k < L; ++k) {
            const float f = (float)(1 << k);
            float sn, cs;
            sincosf(xa * f, &sn, &cs);
            dg[3 + 6 * k + a] = da * f * cs;
            dg[3 + 6 * k + 3 + a] = -da * f * sn;
            if (dx) acc -= da * f * f * (sn * G[(size_t)i * ldg + 3 + 6 * k + a] + cs * G[(size_t)i * ldg + 3 + 6 * k + 3 + a]);
        }
        if (dx) dx[3 * (size_t)i + a] += acc;
    }
}

}  // namespace

#define ST (hipStream_t) stream
extern "C" {

int mp_tr_pe(const float* x, int d_in, int P, int L, int fwd, float scale, float* out, int ld, int col0, void* stream) {
    if (P <= 0) return 0;
    if (d_in == 3) hipLaunchKernelGGL(k_pe_fwd<3>, grid1(P), dim3(TB), 0, ST, x, P, L, fwd, scale, out, ld, col0);
    else if (d_in == 4) hipLaunchKernelGGL(k_pe_fwd<4>, grid1(P), dim3(TB), 0, ST, x, P, L, fwd, scale, out, ld, col0);
    else return -1;
    return (int)hipGetLastError();
}
int mp_tr_softplus_fwd(const float* Z, int ldz, int rows, int C, int P, float scale, float* H, int ldh, int col0,
                       void* stream) {
    if (P <= 0 && mul4(C) && mul4(ldz) && mul4(ldh) && mul4(col0) && al16(Z) && al16(H) && fits32(rows, C))
        hipLaunchKernelGGL(k_softplus_fwd4, grid4(rows, C), dim3(TB), 0, ST, Z, ldz, (unsigned)rows, (unsigned)(C / 4), scale, H, ldh, col0);
    else
        hipLaunchKernelGGL(k_softplus_fwd, grid1((long long)rows * C), dim3(TB), 0, ST, Z, ldz, rows, C, P, scale, H, ldh, col0);
    return (int)hipGetLastError();
}
int mp_tr_softplus_bwd(const float* Z, int ldz, int rows, int C, int P, float scale, const float* dH, int ldh, int col0,
                       float* dZ, int lddz, void* stream) {
    const int vrows = P > 0 ? P : rows;
    hipLaunchKernelGGL(k_softplus_bwd, grid1((long long)vrows * C), dim3(TB), 0, ST, Z, ldz, rows, C, P, scale, dH, ldh, col0,
                       dZ, lddz);
    return (int)hipGetLastError();
}
int mp_tr_relu_bwd(const float* H, int ldh, int rows, int C, const float* dH, int lddh, float* dZ, int lddz, void* stream) {
    if (mul4(C) && mul4(ldh) && mul4(lddh) && mul4(lddz) && al16(H) && al16(dH) && al16(dZ) && fits32(rows, C))
        hipLaunchKernelGGL(k_relu_bwd4, grid4(rows, C), dim3(TB), 0, ST, H, ldh, (unsigned)rows, (unsigned)(C / 4), dH, lddh, dZ, lddz);
    else
        hipLaunchKernelGGL(k_relu_bwd, grid1((long long)rows * C), dim3(TB), 0, ST, H, ldh, (long long)rows * C, C, dH, lddh, dZ,
                           lddz);
    return (int)hipGetLastError();
}
int mp_tr_shade_in_fwd(const float* Z8, int P, int n_pts, const float* xc, const float* jinv, float* XR, float* nrm,
                       float* sdf, const float* grad, void* stream) {
    if (n_pts <= 0) return 0;
    hipLaunchKernelGGL(k_shade_in_fwd, grid1(n_pts), dim3(TB), 0, ST, Z8, P, n_pts, xc, jinv, XR, nrm, sdf, grad);
    return (int)hipGetLastError();
}
int mp_tr_shade_in_bwd(const float* Z8, int P, int n_pts, const float* jinv, const float* dXR, const float* dsdf,
                       const float* dnrm_extra, float* dZ8, float* djinv, const float* grad, float* dgrad, void* stream) {
    if (n_pts <= 0) return 0;
    hipLaunchKernelGGL(k_shade_in_bwd, grid1(n_pts), dim3(TB), 0, ST, Z8, P, n_pts, jinv, dXR, dsdf, dnrm_extra, dZ8, djinv,
                       grad, dgrad);
    return (int)hipGetLastError();
}
int mp_tr_eik_fwd(const float* Z8, int P, int e0, int E, float* grad_theta, const float* grad, void* stream) {
    if (E <= 0) return 0;
    hipLaunchKernelGGL(k_eik_fwd, grid1(E), dim3(TB), 0, ST, Z8, P, e0, E, grad_theta, grad);
    return (int)hipGetLastError();
}
int mp_tr_eik_bwd(int P, int e0, int E, const float* dgrad_theta, float* dZ8, float* dgrad, void* stream) {
    if (E <= 0) return 0;
    hipLaunchKernelGGL(k_eik_bwd, grid1(E), dim3(TB), 0, ST, P, e0, E, dgrad_theta, dZ8, dgrad);
    return (int)hipGetLastError();
}
int mp_tr_sigmoid_fwd(const float* Z, long long n, float* Y, void* stream) {
    hipLaunchKernelGGL(k_sigmoid_fwd, grid1(n), dim3(TB), 0, ST, Z, n, Y);
    return (int)hipGetLastError();
}
int mp_tr_sigmoid_bwd(const float* Y, const float* dY, long long n, float* dZ, void* stream) {
    hipLaunchKernelGGL(k_sigmoid_bwd, grid1(n), dim3(TB), 0, ST, Y, dY, n, dZ);
    return (int)hipGetLastError();
}
int mp_tr_wn_fwd(const float* v, const float* g, int out_dim, int in_dim, float* W, float* WT, void* stream) {
    hipLaunchKernelGGL(k_wn_fwd, dim3(out_dim), dim3(64), 0, ST, v, g, out_dim, in_dim, W, WT);
    return (int)hipGetLastError();
}
int mp_tr_wn_bwd(const float* v, const float* g, int out_dim, int in_dim, const float* dW, float* dv, float* dg,
                 void* stream) {
    hipLaunchKernelGGL(k_wn_bwd, dim3(out_dim), dim3(64), 0, ST, v, g, out_dim, in_dim, dW, dv, dg);
    return (int)hipGetLastError();
}
int mp_tr_wn_fwd_multi(const MpWnDesc* descs, int n_desc, int total_rows, void* stream) {
    if (n_desc <= 0 || total_rows <= 0) return 0;
    hipLaunchKernelGGL(k_wn_fwd_multi, dim3(total_rows), dim3(64), 0, ST, descs, n_desc);
    return (int)hipGetLastError();
}
int mp_tr_wn_bwd_multi(const MpWnDesc* descs, int n_desc, int total_rows, const float* acc_base, float* grad_base, void* stream) {
    if (n_desc <= 0 || total_rows <= 0) return 0;
    hipLaunchKernelGGL(k_wn_bwd_multi, dim3(total_rows), dim3(64), 0, ST, descs, n_desc, acc_base, grad_base);
    return (int)hipGetLastError();
}
int mp_tr_hoist_fwd(const float* W, int out_dim, int in_dim, const float* b, int c0, int n, const float* vec, float* b2,
                    void* stream) {
    hipLaunchKernelGGL(k_hoist_fwd, dim3(out_dim), dim3(64), 0, ST, W, in_dim, b, c0, n, vec, b2);
    return (int)hipGetLastError();
}
int mp_tr_hoist_bwd(const float* db2, int out_dim, int in_dim, int c0, int n, const float* vec, float* dW, void* stream) {
    hipLaunchKernelGGL(k_hoist_bwd, grid1((long long)out_dim * n), dim3(TB), 0, ST, db2, out_dim, in_dim, c0, n, vec, dW);
    return (int)hipGetLastError();
}
int mp_tr_colsum(const float* dZ, int ld, int rows, int C, float* db, void* stream) {
    hipLaunchKernelGGL(k_colsum, dim3(C), dim3(256), 0, ST, dZ, ld, rows, C, db);
    return (int)hipGetLastError();
}
int mp_tr_composite_bwd(int n_rays, int n_person, int n_z, const int* const* inv_index, const float* const* z,
                        const float* const* sdf, const float* const* rgb, const float* beta, const float* bg_rgb,
                        const float* d_rgb_values, const float* d_acc, const float* d_acc_person, float* const* d_sdf,
                        float* const* d_rgb, float* d_bg_rgb, float* d_beta, void* stream) {
    if (n_person > MAX_P) return -1;
    if (n_rays <= 0) return 0;
    const int per_wave = n_person * (5 * (n_z - 1) + 2) * (int)sizeof(float);
    if (per_wave > 160 * 1024) return -2;
    int wpb = 4;
    while (wpb > 1 && wpb * per_wave > 160 * 1024) wpb >>= 1;
    MP_LDS_ATTR((k_composite_bwd), 160 * 1024);
    hipLaunchKernelGGL(k_composite_bwd, dim3((n_rays + wpb - 1) / wpb), dim3(64 * wpb), wpb * per_wave, ST, n_rays, n_person, n_z,
                       inv_index, z, sdf, rgb, beta, bg_rgb, d_rgb_values, d_acc, d_acc_person, d_sdf, d_rgb, d_bg_rgb, d_beta);
    return (int)hipGetLastError();
}
int mp_tr_bg_points(const float* dirs, const float* cam, const float* zbg, int R, int NBG, float radius, float* pts,
                    void* stream) {
    hipLaunchKernelGGL(k_bg_points, grid1((long long)R * NBG), dim3(TB), 0, ST, dirs, cam, zbg, R, NBG, radius, pts);
    return (int)hipGetLastError();
}
int mp_tr_bg_comp_fwd(const float* sdf, const float* rgb, const float* zbg, int R, int NBG, float* out, void* stream) {
    hipLaunchKernelGGL(k_bg_comp_fwd, grid1(R), dim3(TB), 0, ST, sdf, rgb, zbg, R, NBG, out);
    return (int)hipGetLastError();
}
int mp_tr_bg_comp_bwd(const float* sdf, const float* rgb, const float* zbg, int R, int NBG, const float* dout, float* dsdf,
                      float* drgb, void* stream) {
    hipLaunchKernelGGL(k_bg_comp_bwd, grid1(R), dim3(TB), 0, ST, sdf, rgb, zbg, R, NBG, dout, dsdf, drgb);
    return (int)hipGetLastError();
}
int mp_tr_pe_bwd(const float* x, int d_in, int P, int L, int fwd, const float* dIN, int ld, float* dx, void* stream) {
    if (P <= 0) return 0;
    if (d_in == 3) hipLaunchKernelGGL(k_pe_bwd<3>, grid1(P), dim3(TB), 0, ST, x, P, L, fwd, dIN, ld, dx);
    else if (d_in == 4) hipLaunchKernelGGL(k_pe_bwd<4>, grid1(P), dim3(TB), 0, ST, x, P, L, fwd, dIN, ld, dx);
    else return -1;
    return (int)hipGetLastError();
}
int mp_tr_warp_bwd(const float* xc, const float* dxc, const float* jinv, const float* djinv, const int* nn_posed,
                   const int* nn_cano, int n, const float* skin_w, const float* tfs, float* dtfs, void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_warp_bwd, grid1(n), dim3(TB), 0, ST, xc, dxc, jinv, djinv, nn_posed, nn_cano, n, skin_w, tfs, dtfs);
    return (int)hipGetLastError();
}
int mp_smpl_pose_bwd(const int* parents, const float* params, const float* tfs_c_inv, const float* rest_joints,
                     const float* j_shapedirs, const float* dtfs, float* dparams, void* stream) {
    hipLaunchKernelGGL(k_smpl_pose_bwd, dim3(1), dim3(64), 0, ST, parents, params, tfs_c_inv, rest_joints, j_shapedirs, dtfs,
                       dparams);
    return (int)hipGetLastError();
}
int mp_tr_sigmul(const float* Z, int ldz, long long rows, int C, const float* U, int ldu, const float* wrow, float scale,
                 float* V, int ldv, void* stream) {
    if (mul4(C) && mul4(ldz) && mul4(ldv) && (U ? mul4(ldu) && al16(U) : al16(wrow)) && al16(Z) && al16(V) && fits32(rows, C))
        hipLaunchKernelGGL(k_sigmul4, grid4(rows, C), dim3(TB), 0, ST, Z, ldz, (unsigned)rows, (unsigned)(C / 4), U, ldu, wrow, scale, V, ldv);
    else
        hipLaunchKernelGGL(k_sigmul, grid1(rows * C), dim3(TB), 0, ST, Z, ldz, rows, C, U, ldu, wrow, scale, V, ldv);
    return (int)hipGetLastError();
}
int mp_tr_rev_adj(const float* Z, int ldz, long long rows, int C, const float* U, int ldu, const float* wrow, float uscale,
                  const float* dV, int lddv, float* dU, int lddu, float* dS, int ldds, void* stream) {
    if (mul4(C) && mul4(ldz) && mul4(lddv) && mul4(lddu) && mul4(ldds) && (U ? mul4(ldu) && al16(U) : al16(wrow)) && al16(Z) &&
        al16(dV) && al16(dU) && al16(dS) && fits32(rows, C))
        hipLaunchKernelGGL(k_rev_adj4, grid4(rows, C), dim3(TB), 0, ST, Z, ldz, (unsigned)rows, (unsigned)(C / 4), U, ldu, wrow, uscale, dV,
                           lddv, dU, lddu, dS, ldds);
    else
        hipLaunchKernelGGL(k_rev_adj, grid1(rows * C), dim3(TB), 0, ST, Z, ldz, rows, C, U, ldu, wrow, uscale, dV, lddv, dU, lddu, dS,
                           ldds);
    return (int)hipGetLastError();
}
int mp_tr_dz(const float* Z, int ldz, long long rows, int C, const float* dX, int lddx, float scale, const float* dS, int ldds,
             float* dZ, int lddz, void* stream) {
    if (mul4(C) && mul4(ldz) && mul4(lddx) && mul4(ldds) && mul4(lddz) && al16(Z) && al16(dX) && al16(dS) && al16(dZ) && fits32(rows, C))
        hipLaunchKernelGGL(k_dz4, grid4(rows, C), dim3(TB), 0, ST, Z, ldz, (unsigned)rows, (unsigned)(C / 4), dX, lddx, scale, dS, ldds, dZ,
                           lddz);
    else
        hipLaunchKernelGGL(k_dz, grid1(rows * C), dim3(TB), 0, ST, Z, ldz, rows, C, dX, lddx, scale, dS, ldds, dZ, lddz);
    return (int)hipGetLastError();
}
int mp_tr_pe_grad_fwd(const float* x, int P, int L, const float* G, int ldg, float* grad, void* stream) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(k_pe_grad_fwd, grid1(P), dim3(TB), 0, ST, x, P, L, G, ldg, grad);
    return (int)hipGetLastError();
}
int mp_tr_pe_grad_bwd(const float* x, int P, int L, const float* dgrad, const float* G, int ldg, float* dG, int lddg,
                      float* dx, void* stream) {
    if (P <= 0) return 0;
    hipLaunchKernelGGL(k_pe_grad_bwd, grid1(P), dim3(TB), 0, ST, x, P, L, dgrad, G, ldg, dG, lddg, dx);
    return (int)hipGetLastError();
}
int mp_tr_copy_cols(const float* src, int lds, int c0s, float* dst, int ldd, int c0d, long long rows, int C, float scale,
                    int accumulate, void* stream) {
    hipLaunchKernelGGL(k_copy_cols, grid1(rows * C), dim3(TB), 0, ST, src, lds, c0s, dst, ldd, c0d, rows, C, scale, accumulate);
    return (int)hipGetLastError();
}
}
