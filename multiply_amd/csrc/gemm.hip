// fp32 GEMMs on the exact-f32 matrix cores (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate, = the fp32 vector rate)
// for the TRAINING path: per-layer products of the scene MLPs over the ~10^5..10^6 sample rows of one 512-ray iteration.
// The training path keeps the reference's fp32 arithmetic (the reference trains in fp32, no autocast), so gradients
// can be compared with torch autograd tightly; the bf16 fused kernels (mlp.hip) stay the inference / sampler path.
//   mp_gemm_nt : C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias[N] on the first bias_rows rows) (optionally ReLU)
//   mp_gemm_tn : C[M,N] += A[K,M]^T . B[K,N]   (contraction over the ROW index, split over blocks, fp32 atomics)
//                optionally colsum[m] += sum_{r < colsum_rows} A[r][m]  (the bias gradient rides on the same A tiles)
//
// Both: 128x128 tile of C per workgroup (8 waves, each 64x32 = 4x2 MFMA blocks; two workgroups per CU = 4 waves per SIMD,
// which is what hides the LDS / barrier latencies: the 4-wave version ran at 73 TFLOP/s, this one at 86), depth 32 per
// LDS stage, two stages:
// the global loads of stage t+1 (float4 per lane, coalesced along the contiguous dimension) are in flight while the
// MFMAs of stage t run; one barrier per stage.  The MFMA's k index is free to permute (a sum), and so is the mapping
// of a lane's row/column inside the tile, so both are chosen such that every operand fetch is one ds_read_b128:
//   nt: lane (li, lq) takes k = 8 lq + s for the 8 MFMA steps s of a stage  -> 8 consecutive floats of its row
//   tn: lane li of row block i owns tile row 4 li + i (column 2 li + j)      -> 4 (2) consecutive floats of LDS row r
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "../../include/multiply_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NT_THREADS = 512;   // nt kernel: 8 waves (2 x 4), each a 64 x 32 sub-tile -> 4 waves per SIMD at 2 workgroups per CU
constexpr int LDK = BK + 4;    // nt: [row][k] tiles, row stride 36 floats (16 B aligned, b128 reads spread over all banks)
constexpr int LDM = BM + 4;    // tn: [r][m] tiles

__device__ __forceinline__ f32x4 ld4(const float* p, bool ok4, int n_valid) {
    // n_valid: how many of the 4 elements are inside the matrix (<= 0: none).  ok4: all four, and 16 B aligned
    if (ok4) return *(const f32x4*)p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int e = 0; e < 4; ++e)
        if (e < n_valid) v[e] = p[e];
    return v;
}

// FAST: 16 B aligned operands, leading dimensions and K multiples of 4 / BK: the k loop carries no bounds logic at all
// (rows past M / N are clamped to the last valid row -- their products land in C entries that are never written).
template <bool FAST>
__global__ __launch_bounds__(NT_THREADS) void k_gemm_nt(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 float* __restrict__ C, int ldc, int M, int N, int K,
                                                 const float* __restrict__ bias, int bias_rows, int accumulate, int relu) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*As)[BM * LDK] = (float (*)[BM * LDK])smem;
    float (*Bs)[BN * LDK] = (float (*)[BN * LDK])(smem + 2 * BM * LDK);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;   // 2 x 4 waves, each 64 x 32
    const int li = lane & 15, lq = lane >> 4;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    // staging: 128 rows x 8 float4 per tile = 1024 float4 / 512 threads = 2 per thread per operand
    const int sr = t >> 3, sk = (t & 7) * 4;       // row sr + 64 q, k offset sk
    const bool a_al = (lda & 3) == 0 && ((size_t)A & 15) == 0, b_al = (ldb & 3) == 0 && ((size_t)B & 15) == 0;
    f32x4 ra[2], rb[2];
    const float* pa[2];
    const float* pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pa[q] = A + (size_t)min(m0 + sr + 64 * q, M - 1) * lda + sk;
        pb[q] = B + (size_t)min(n0 + sr + 64 * q, N - 1) * ldb + sk;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if constexpr (FAST) {
                ra[q] = *(const f32x4*)(pa[q] + k0);
                rb[q] = *(const f32x4*)(pb[q] + k0);
            } else {
                const int am = m0 + sr + 64 * q, bn = n0 + sr + 64 * q, k = k0 + sk;
                ra[q] = am < M ? ld4(A + (size_t)am * lda + k, a_al && k + 3 < K, K - k) : (f32x4){0, 0, 0, 0};
                rb[q] = bn < N ? ld4(B + (size_t)bn * ldb + k, b_al && k + 3 < K, K - k) : (f32x4){0, 0, 0, 0};
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *(f32x4*)&As[buf][(sr + 64 * q) * LDK + sk] = ra[q];
            *(f32x4*)&Bs[buf][(sr + 64 * q) * LDK + sk] = rb[q];
        }
    };
    const int nk = (K + BK - 1) / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        f32x4 a[4][2], b[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i][0] = *(const f32x4*)&As[buf][(wm + i * 16 + li) * LDK + 8 * lq];
            a[i][1] = *(const f32x4*)&As[buf][(wm + i * 16 + li) * LDK + 8 * lq + 4];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            b[j][0] = *(const f32x4*)&Bs[buf][(wn + j * 16 + li) * LDK + 8 * lq];
            b[j][1] = *(const f32x4*)&Bs[buf][(wn + j * 16 + li) * LDK + 8 * lq + 4];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s >> 2][s & 3], b[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    // D: col = lane&15 (n), row = 4*(lane>>4)+reg (m)
    const int cn = lane & 15, cr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + j * 16 + cn;
            if (n >= N) continue;
            const float bn = bias ? bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + cr + r;
                if (m >= M) continue;
                float v = acc[i][j][r];
                if (m < bias_rows) v += bn;
                float* c = C + (size_t)m * ldc + n;
                if (accumulate) v += *c;
                if (relu) v = fmaxf(v, 0.0f);
                *c = v;
            }
        }
}

// C[M,N] += sum_r A[r,m] B[r,n]; block = 128x128 tile of C x one slice of the rows
// FAST: aligned operands, M and N multiples of the tile: only the row-slice bound remains in the loop
template <bool FAST>
__global__ __launch_bounds__(NT_THREADS) void k_gemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 float* __restrict__ C, int ldc, int M, int N, int K, int rows_per_block,
                                                 float* __restrict__ colsum, int colsum_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*As)[BK * LDM] = (float (*)[BK * LDM])smem;
    float (*Bs)[BK * LDM] = (float (*)[BK * LDM])(smem + 2 * BK * LDM);
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int r_begin = blockIdx.z * rows_per_block, r_end = min(K, r_begin + rows_per_block);
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;   // 2 x 4 waves, each 64 x 32
    const int li = lane & 15, lq = lane >> 4;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    // staging: 32 rows x 32 float4 per tile = 1024 float4 / 512 threads = 2 per thread per operand
    const int sc = (t & 31) * 4, sr = t >> 5;      // columns sc..sc+3, rows sr + 16 q
    const bool a_al = (lda & 3) == 0 && ((size_t)A & 15) == 0, b_al = (ldb & 3) == 0 && ((size_t)B & 15) == 0;
    const bool do_sum = colsum != nullptr && blockIdx.y == 0;
    f32x4 ra[2], rb[2], csum = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int r0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = r0 + sr + 16 * q;
            const int am = m0 + sc, bn = n0 + sc;
            if constexpr (FAST) {
                ra[q] = r < r_end ? *(const f32x4*)(A + (size_t)r * lda + am) : (f32x4){0, 0, 0, 0};
                rb[q] = r < r_end ? *(const f32x4*)(B + (size_t)r * ldb + bn) : (f32x4){0, 0, 0, 0};
            } else {
                ra[q] = r < r_end ? ld4(A + (size_t)r * lda + am, a_al && am + 3 < M, M - am) : (f32x4){0, 0, 0, 0};
                rb[q] = r < r_end ? ld4(B + (size_t)r * ldb + bn, b_al && bn + 3 < N, N - bn) : (f32x4){0, 0, 0, 0};
            }
            if (do_sum && r < colsum_rows) csum += ra[q];
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *(f32x4*)&As[buf][(sr + 16 * q) * LDM + sc] = ra[q];
            *(f32x4*)&Bs[buf][(sr + 16 * q) * LDM + sc] = rb[q];
        }
    };
    const int nk = (r_end - r_begin + BK - 1) / BK;
    if (nk > 0) {
        gload(r_begin);
        sstore(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(r_begin + (kt + 1) * BK);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // lane (li, lq) contracts LDS row r = 4 s + lq; its tile rows are 4 li + i (i = 0..3), its columns 2 li + j
            const f32x4 a = *(const f32x4*)&As[buf][(4 * s + lq) * LDM + wm + 4 * li];
            const f32x2 b = *(const f32x2*)&Bs[buf][(4 * s + lq) * LDM + wn + 2 * li];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }
    // D block (i, j): lane holds D[row 4*(lane>>4)+reg][col lane&15] of the MFMA block = tile row wm + 4*(4*(lane>>4)+reg) + i,
    // tile column wn + 2*(lane&15) + j
    const int dc = lane & 15, dr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 2 * dc + j;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 4 * (dr + r) + i;
                if (m < M) atomicAdd(C + (size_t)m * ldc + n, acc[i][j][r]);
            }
        }
    if (do_sum) {   // threads with the same sc hold partial sums of the same 4 columns: fold them through LDS
        __syncthreads();
        float* red = smem;
        for (int i = t; i < BM; i += NT_THREADS) red[i] = 0.f;
        __syncthreads();
        for (int e = 0; e < 4; ++e) atomicAdd(&red[sc + e], csum[e]);
        __syncthreads();
        for (int i = t; i < BM; i += NT_THREADS)
            if (m0 + i < M && red[i] != 0.f) atomicAdd(colsum + m0 + i, red[i]);
    }
}

}  // namespace

extern "C" int mp_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                          const float* bias, int bias_rows, int accumulate, int relu, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    constexpr int LDS_NT = 2 * (BM + BN) * LDK * (int)sizeof(float);
    MP_LDS_ATTR((k_gemm_nt<true>), LDS_NT);
    MP_LDS_ATTR((k_gemm_nt<false>), LDS_NT);
    const bool fast = (lda & 3) == 0 && (ldb & 3) == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0 && K % BK == 0;
    if (fast)
        hipLaunchKernelGGL(k_gemm_nt<true>, dim3((M + BM - 1) / BM, (N + BN - 1) / BN), dim3(NT_THREADS), LDS_NT, (hipStream_t)stream,
                           A, lda, B, ldb, C, ldc, M, N, K, bias, bias_rows, accumulate, relu);
    else
        hipLaunchKernelGGL(k_gemm_nt<false>, dim3((M + BM - 1) / BM, (N + BN - 1) / BN), dim3(NT_THREADS), LDS_NT, (hipStream_t)stream,
                           A, lda, B, ldb, C, ldc, M, N, K, bias, bias_rows, accumulate, relu);
    return (int)hipGetLastError();
}

extern "C" int mp_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                          float* colsum, int colsum_rows, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // slices of the contraction: enough workgroups to fill 256 CUs twice, at least 4 stages (128 rows) each
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int slices = (512 + tiles - 1) / tiles;
    int rows = (K + slices - 1) / slices;
    rows = (rows + BK - 1) / BK * BK;
    if (rows < 4 * BK) rows = 4 * BK;
    slices = (K + rows - 1) / rows;
    constexpr int LDS_TN = 4 * BK * LDM * (int)sizeof(float);
    MP_LDS_ATTR((k_gemm_tn<true>), LDS_TN);
    MP_LDS_ATTR((k_gemm_tn<false>), LDS_TN);
    const bool fast = (lda & 3) == 0 && (ldb & 3) == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0 && M % BM == 0 &&
                      N % BN == 0;
    if (fast)
        hipLaunchKernelGGL(k_gemm_tn<true>, dim3(M / BM, N / BN, slices), dim3(NT_THREADS), LDS_TN, (hipStream_t)stream, A, lda, B, ldb, C,
                           ldc, M, N, K, rows, colsum, colsum_rows);
    else
        hipLaunchKernelGGL(k_gemm_tn<false>, dim3((M + BM - 1) / BM, (N + BN - 1) / BN, slices), dim3(NT_THREADS), LDS_TN,
                           (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, rows, colsum, colsum_rows);
    return (int)hipGetLastError();
}
