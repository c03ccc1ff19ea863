// fp32 GEMMs on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate, = the fp32 vector rate)
// for the TRAINING path: per-layer products of the scene MLPs over the ~10^5 sample points of one 512-ray iteration.
// The training path keeps the reference's fp32 arithmetic (the reference trains in fp32, no autocast), so gradients
// can be compared with torch autograd tightly; the bf16 fused kernels (mlp.hip) stay the inference / sampler path.
//   mp_gemm_nt : C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N] on the first bias_rows rows) (optionally ReLU)
//   mp_gemm_tn : C[M,N] += A[K,M]^T . B[K,N]   (contraction over the ROW index, split over blocks, fp32 atomics)
#include <hip/hip_runtime.h>
#include "../../include/multiply_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = BK + 1;

// C tile 128x128 per block (4 waves, each 64x64 = 4x4 MFMA blocks), K in steps of 16
__global__ __launch_bounds__(256) void k_gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 float* __restrict__ C, int ldc, int M, int N, int K,
                                                 const float* __restrict__ bias, int bias_rows, int accumulate, int relu) {
    __shared__ float As[BM * LDT];
    __shared__ float Bs[BN * LDT];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const int lr = t >> 1, lk = (t & 1) * 8;  // this thread stages row lr, k-offset lk..lk+7 of both tiles
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + lk + e;
            const int am = m0 + lr, bn = n0 + lr;
            As[lr * LDT + lk + e] = (am < M && k < K) ? A[(size_t)am * lda + k] : 0.0f;
            Bs[lr * LDT + lk + e] = (bn < N && k < K) ? B[(size_t)bn * ldb + k] : 0.0f;
        }
        __syncthreads();
        const int li = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(wm + i * 16 + li) * LDT + kk + lq];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(wn + j * 16 + li) * LDT + kk + lq];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // D: col = lane&15 (n), row = 4*(lane>>4)+reg (m)
    const int cn = lane & 15, cr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn + j * 16 + cn;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + cr + r;
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (bias && m < bias_rows) v += bias[n];
                float* c = C + (size_t)m * ldc + n;
                if (accumulate) v += *c;
                if (relu) v = fmaxf(v, 0.0f);
                *c = v;
            }
        }
}

// C[M,N] += sum_r A[r,m] B[r,n]; block = 64x64 tile of C x one slice of the rows; 4 waves each 32x32 (2x2 MFMA blocks)
constexpr int TM = 64, TN = 64, TK = 16;
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 float* __restrict__ C, int ldc, int M, int N, int K, int rows_per_block) {
    __shared__ float As[TK * TM];
    __shared__ float Bs[TK * TN];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
    const int r_begin = blockIdx.z * rows_per_block, r_end = min(K, r_begin + rows_per_block);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const int sr = t >> 4, sc = (t & 15) * 4;  // stage row sr (0..15), columns sc..sc+3
    for (int r0 = r_begin; r0 < r_end; r0 += TK) {
        const int r = r0 + sr;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[sr * TM + sc + e] = (r < r_end && m0 + sc + e < M) ? A[(size_t)r * lda + m0 + sc + e] : 0.0f;
            Bs[sr * TN + sc + e] = (r < r_end && n0 + sc + e < N) ? B[(size_t)r * ldb + n0 + sc + e] : 0.0f;
        }
        __syncthreads();
        const int li = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < TK; kk += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(kk + lq) * TM + wm + i * 16 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + lq) * TN + wn + j * 16 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int cn = lane & 15, cr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 16 + cn;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + cr + r;
                if (m < M) atomicAdd(C + (size_t)m * ldc + n, acc[i][j][r]);
            }
        }
}

}  // namespace

extern "C" int mp_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                          const float* bias, int bias_rows, int accumulate, int relu, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    hipLaunchKernelGGL(k_gemm_nt, dim3((M + BM - 1) / BM, (N + BN - 1) / BN), dim3(256), 0, (hipStream_t)stream, A, lda, B,
                       ldb, C, ldc, M, N, K, bias, bias_rows, accumulate, relu);
    return (int)hipGetLastError();
}

extern "C" int mp_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                          void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int slices = (K + 2047) / 2048;
    if (slices > 1024) slices = 1024;
    int rows = (K + slices - 1) / slices;
    rows = (rows + TK - 1) / TK * TK;
    slices = (K + rows - 1) / rows;
    hipLaunchKernelGGL(k_gemm_tn, dim3((M + TM - 1) / TM, (N + TN - 1) / TN, slices), dim3(256), 0, (hipStream_t)stream, A,
                       lda, B, ldb, C, ldc, M, N, K, rows);
    return (int)hipGetLastError();
}
