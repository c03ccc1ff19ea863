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

// ---- mp_gemm_nt_bf16x3: the same product on the 16-bit matrix cores at (almost) fp32 accuracy ---------------------------------
// Every fp32 operand is split on its way into LDS into two bfloat16 halves, x = hi + lo + r with |r| <= 2^-17 |x| (hi = bf16(x),
// lo = bf16(x - hi)), and a product is three MFMAs, hi.hi + hi.lo + lo.hi (the dropped lo.lo term is 2^-16 relative): relative
// error ~2^-16 per product instead of the 2^-9 of a plain bf16 product, the RANGE of fp32 (no loss scaling: the gradient
// operands of the backward pass reach 1e-9, below half precision's subnormals), fp32 accumulation.  One v_mfma_f32_16x16x32_bf16
// (16 cycles) covers the whole 32-deep LDS stage of a 16x16 block, so a stage costs 8 blocks x 3 = 24 MFMAs = 384 matrix-pipe
// cycles per wave against 64 x 32 = 2048 with v_mfma_f32_16x16x4_f32: the training GEMMs (M ~ 5e4 rows, N = K = 256) leave the
// compute-bound regime and run at the rate HBM delivers their fp32 operands.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// LDS rows of the split tiles.  MP_NT_SWZ = 0: 80-byte rows (32 bf16 + 8 of padding) -- laid out for sixteen CONSECUTIVE lanes per
// LDS cycle, but a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): every group
// then has a 2-way bank conflict.  MP_NT_SWZ = 1: unpadded 64-byte rows whose four 16-byte chunks are stored at chunk ^ ((row >> 1)
// & 3): conflict-free under the real groups for the fragment reads and for the staging threads' 8-byte stores, and 64 instead of
// 80 KB per workgroup.
#ifndef MP_NT_SWZ
#define MP_NT_SWZ 0
#endif
constexpr int LDK16 = MP_NT_SWZ ? BK : BK + 8;   // bf16 elements per LDS row
__device__ __forceinline__ int swz16(int row, int chunk) { return MP_NT_SWZ ? chunk ^ ((row >> 1) & 3) : chunk; }

// PA: how many stages ahead the A operand (the activation rows: HBM) is fetched through a register ring (the k loop is
// unrolled by PA, so the ring slots are compile-time).  Measured on a 50k x 256 x 256 product: PA = 1 (the shape of the fp32
// kernel) 44.8 us, PA = 4 49.4 us; a 128 x 256 tile per workgroup (activation rows read once instead of once per column tile,
// 120 KB of LDS = one workgroup per CU) 58.6 us -- neither memory latency nor the repeated read is what bounds it (the repeat is
// served by the Infinity Cache); at 100-150 MB per product it runs at 2.2-3.3 TB/s.  PA = 1 is what is instantiated.
// Fragments of A are read per row block inside the MFMA loop: 91 VGPRs, two workgroups (four waves per SIMD) per CU.
// NJ: 16-column blocks per wave (2: a 128 x 128 tile of C per workgroup; 4: 128 x 256 = the whole width of the 256-wide layers,
// the activation rows are fetched ONCE instead of once per column tile);  NST: LDS stages (2: loads of stage t + 1 stored while
// stage t is multiplied, one barrier per stage; 1: one buffer, two barriers per stage, half the LDS -> twice the workgroups per CU)
template <bool FAST, int PA, int NJ, int NST>
__global__ __launch_bounds__(NT_THREADS, 2) void k_gemm_nt_b3(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                       float* __restrict__ C, int ldc, int M, int N, int K,
                                                       const float* __restrict__ bias, int bias_rows, int accumulate, int relu) {
    static_assert(PA == 1 || FAST, "the deep prefetch is for the bounds-free path");
    constexpr int BNW = 64 * NJ, QB = BNW / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [stage][hi | lo][row][k] for A, then the same for B
    __bf16* As = (__bf16*)smem;
    __bf16* Bs = As + NST * 2 * BM * LDK16;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BNW;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 16 * NJ;   // 2 x 4 waves, each 64 x 16 NJ
    const int li = lane & 15, lq = lane >> 4;
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    const int sr = t >> 3, sk = (t & 7) * 4;       // staging: row sr + 64 q, k offset sk
    const bool a_al = (lda & 3) == 0 && ((size_t)A & 15) == 0, b_al = (ldb & 3) == 0 && ((size_t)B & 15) == 0;
    f32x4 ra[PA][2], rb[QB];
    const float* pa[2];
    const float* pb[QB];
#pragma unroll
    for (int q = 0; q < 2; ++q) pa[q] = A + (size_t)min(m0 + sr + 64 * q, M - 1) * lda + sk;
#pragma unroll
    for (int q = 0; q < QB; ++q) pb[q] = B + (size_t)min(n0 + sr + 64 * q, N - 1) * ldb + sk;
    auto gload_a = [&](f32x4 (&r)[2], int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if constexpr (FAST) {
                r[q] = *(const f32x4*)(pa[q] + k0);
            } else {
                const int am = m0 + sr + 64 * q, k = k0 + sk;
                r[q] = am < M ? ld4(A + (size_t)am * lda + k, a_al && k + 3 < K, K - k) : (f32x4){0, 0, 0, 0};
            }
        }
    };
    auto gload_b = [&](int k0) {
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if constexpr (FAST) {
                rb[q] = *(const f32x4*)(pb[q] + k0);
            } else {
                const int bn = n0 + sr + 64 * q, k = k0 + sk;
                rb[q] = bn < N ? ld4(B + (size_t)bn * ldb + k, b_al && k + 3 < K, K - k) : (f32x4){0, 0, 0, 0};
            }
        }
    };
    auto split_store = [&](__bf16* tile, int rows, f32x4 v, int row) {   // tile = [hi | lo][rows][k] of one stage
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
        const int ks = 8 * swz16(row, sk >> 3) + (sk & 4);
        *(bf16x4*)(tile + row * LDK16 + ks) = hi;
        *(bf16x4*)(tile + rows * LDK16 + row * LDK16 + ks) = lo;
    };
    auto sstore = [&](int buf, const f32x4 (&r)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) split_store(As + buf * 2 * BM * LDK16, BM, r[q], sr + 64 * q);
#pragma unroll
        for (int q = 0; q < QB; ++q) split_store(Bs + buf * 2 * BNW * LDK16, BNW, rb[q], sr + 64 * q);
    };
    auto mfma_stage = [&](int buf) {
        const __bf16* at = As + buf * 2 * BM * LDK16;
        const __bf16* bt = Bs + buf * 2 * BNW * LDK16;
        bf16x8 bh[NJ], bl[NJ];
        // lane (li, lq) supplies k = 8 lq .. 8 lq + 7 of its row (the k index of an MFMA is free to permute): one b128 per operand
        const int lqs = 8 * swz16(li, lq);      // the tile rows of a lane are li + multiples of 16: the swizzle depends on li only
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[j] = *(const bf16x8*)(bt + (wn + j * 16 + li) * LDK16 + lqs);
            bl[j] = *(const bf16x8*)(bt + BNW * LDK16 + (wn + j * 16 + li) * LDK16 + lqs);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 ah = *(const bf16x8*)(at + (wm + i * 16 + li) * LDK16 + lqs);
            const bf16x8 al = *(const bf16x8*)(at + BM * LDK16 + (wm + i * 16 + li) * LDK16 + lqs);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);   // the small terms first
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    const int nk = (K + BK - 1) / BK;
#pragma unroll
    for (int s_ = 0; s_ < PA; ++s_)
        if (s_ < nk) gload_a(ra[s_], s_ * BK);
    gload_b(0);
    sstore(0, ra[0]);
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PA) {
#pragma unroll
        for (int u = 0; u < PA; ++u) {
            const int kt = kt0 + u, buf = NST == 2 ? (kt & 1) : 0;
            if (kt < nk) {
                if (kt + 1 < nk) gload_b((kt + 1) * BK);
                if (kt + PA < nk) gload_a(ra[u], (kt + PA) * BK);     // slot u held stage kt: stored to LDS one iteration ago
                mfma_stage(buf);
                if constexpr (NST == 1) __syncthreads();               // every wave is done reading the only buffer
                if (kt + 1 < nk) sstore(NST == 2 ? (buf ^ 1) : 0, ra[(u + 1) % PA]);
                __syncthreads();
            }
        }
    }
    // D: col = lane&15 (n), row = 4*(lane>>4)+reg (m)
    const int cn = lane & 15, cr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
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

// ---- the same product for the 256-deep layers as a PERSISTENT workgroup (K == 256, 16 B aligned operands) ------------------
// The tiled kernel above runs the 63k x 256 x 256 products of a training iteration in 45 us: ~20 us that do not depend on K
// (first-load latency and epilogue of every tile, exposed with two workgroups per CU) + an LDS-bound K loop (a 64 x 32 wave
// tile reads 12 operand fragments per 24 MFMAs; profiles/r03_gemm_shape_sweep.txt).  Here one workgroup per CU keeps the
// weights of its 128 columns IN REGISTERS for the whole launch -- lane (li, lq) of a wave holds, for its 2 column blocks and
// the 8 stages, the hi and lo halves of 8 consecutive k of its column: 8 x 2 x 2 x 4 = 128 VGPRs, loaded and split once --
// and walks the row tiles blockIdx.x, blockIdx.x + gridDim.x, ...: only the activation rows stream, fetched FOUR stages ahead
// through a register ring ACROSS tile boundaries (the next tile's first stages are in flight during the epilogue), split into
// the two-stage LDS tile as above.  Per stage a wave reads 8 fragments (A only) for 24 MFMAs: the loop is MFMA-bound.
constexpr int NKP = 8;            // stages of the 256-deep contraction
__global__ __launch_bounds__(NT_THREADS, 1) void k_gemm_nt_b3p(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                        int ldb, float* __restrict__ C, int ldc, int M, int N,
                                                        const float* __restrict__ bias, int bias_rows, int relu, int row_tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* As = (__bf16*)smem;                              // [stage][hi | lo][row][k]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int n0 = blockIdx.y * 128;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;   // 2 x 4 waves, each 64 x 32
    const int li = lane & 15, lq = lane >> 4;
    // the wave's weights: column n0 + wn + 16 j + li, k = 32 kt + 8 lq .. + 7 (columns past N: the last valid one, never stored)
    bf16x8 bh[NKP][2], bl[NKP][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float* pb = B + (size_t)min(n0 + wn + j * 16 + li, N - 1) * ldb + 8 * lq;
#pragma unroll
        for (int kt = 0; kt < NKP; ++kt) {
            const f32x4 v0 = *(const f32x4*)(pb + kt * BK), v1 = *(const f32x4*)(pb + kt * BK + 4);
            const bf16x4 h0 = __builtin_convertvector(v0, bf16x4), h1 = __builtin_convertvector(v1, bf16x4);
            const bf16x4 l0 = __builtin_convertvector(v0 - __builtin_convertvector(h0, f32x4), bf16x4);
            const bf16x4 l1 = __builtin_convertvector(v1 - __builtin_convertvector(h1, f32x4), bf16x4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bh[kt][j][e] = h0[e]; bh[kt][j][4 + e] = h1[e];
                bl[kt][j][e] = l0[e]; bl[kt][j][4 + e] = l1[e];
            }
        }
    }
    const int sr = t >> 3, sk = (t & 7) * 4;                 // staging: row sr + 64 q, k offset sk
    auto split_store = [&](__bf16* tile, f32x4 v, int row) {
        const bf16x4 hi = __builtin_convertvector(v, bf16x4);
        const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
        const int ks = 8 * swz16(row, sk >> 3) + (sk & 4);
        *(bf16x4*)(tile + row * LDK16 + ks) = hi;
        *(bf16x4*)(tile + BM * LDK16 + row * LDK16 + ks) = lo;
    };
    auto sstore = [&](int buf, const f32x4 (&r)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) split_store(As + buf * 2 * BM * LDK16, r[q], sr + 64 * q);
    };
    // element offsets of the staging thread's two rows (kept as offsets from the kernel argument: pointers copied between
    // tiles lose their address space and become flat loads, which every LDS wait would then also wait for)
    auto tile_offs = [&](int tile, size_t (&o)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) o[q] = (size_t)min(tile * BM + sr + 64 * q, M - 1) * lda + sk;
    };
    // The loop below holds NO conditional memory operation (the next tile's rows are fetched even when there is none: the last
    // tile's again; no accumulate path; the bias is read here): hipcc's s_waitcnt bookkeeping merges the outstanding-load counts
    // of all paths into the smallest, and with `if (has_next)` loads it waited for all but one load at every stage.
    const int GX = gridDim.x;
    if ((int)blockIdx.x >= row_tiles) return;
    const int ntile = (row_tiles - (int)blockIdx.x + GX - 1) / GX;
    const int cn = lane & 15, cr = (lane >> 4) * 4;
    float bn[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bn[j] = bias ? bias[min(n0 + wn + j * 16 + cn, N - 1)] : 0.0f;
    int tile = blockIdx.x;
    size_t pc[2], pn[2];                                     // this tile's rows, the next tile's rows
    tile_offs(tile, pc);
    tile_offs(ntile > 1 ? tile + GX : tile, pn);
    f32x4 ra[4][2];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int q = 0; q < 2; ++q) ra[s_][q] = *(const f32x4*)(A + pc[q] + s_ * BK);
    sstore(0, ra[0]);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int kt = 0; kt < NKP; ++kt) {
            const int buf = kt & 1;
            // stage kt + 4 into the ring slot stage kt left (its rows went to LDS one stage ago): this tile's, or the next tile's
#if !(MP_EXP_NTP & 4)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                ra[kt & 3][q] = kt + 4 < NKP ? *(const f32x4*)(A + pc[q] + (kt + 4) * BK) : *(const f32x4*)(A + pn[q] + (kt + 4 - NKP) * BK);
#endif
            __builtin_amdgcn_sched_barrier(0);
            const __bf16* at = As + buf * 2 * BM * LDK16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 ah = *(const bf16x8*)(at + (wm + i * 16 + li) * LDK16 + 8 * swz16(li, lq));
                const bf16x8 al = *(const bf16x8*)(at + BM * LDK16 + (wm + i * 16 + li) * LDK16 + 8 * swz16(li, lq));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#if MP_EXP_NTP & 2
                    acc[i][j][0] += (float)al[0] + (float)ah[1] + (float)bh[kt][j][0] + (float)bl[kt][j][1];
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[kt][j], acc[i][j], 0, 0, 0);   // the small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[kt][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[kt][j], acc[i][j], 0, 0, 0);
#endif
                }
            }
            sstore(buf ^ 1, ra[(kt + 1) & 3]);               // stage kt + 1 (kt = 7: the next tile's first)
            __syncthreads();
        }
        // D: col = lane&15 (n), row = 4*(lane>>4)+reg (m)
        const int m0 = tile * BM;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn + j * 16 + cn;
                if (n >= N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm + i * 16 + cr + r;
                    if (m >= M) continue;
                    float v = acc[i][j][r];
                    if (m < bias_rows) v += bn[j];
                    if (relu) v = fmaxf(v, 0.0f);
#if MP_EXP_NTP & 1
                    if (v == 123.456f)
#endif
                    C[(size_t)m * ldc + n] = v;
                }
            }
        tile += GX;
        pc[0] = pn[0]; pc[1] = pn[1];
        tile_offs(it + 2 < ntile ? tile + GX : tile, pn);
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

// ---- mp_gemm_tn_bf16x3: the weight-gradient contraction with the same split -------------------------------------------------
// C[m][n] += sum_r A[r][m] B[r][n]: the MFMA's k index is the ROW index of both operands, which are row-major in HBM -- each lane
// needs consecutive rows of one column.  The transposition happens in registers on the way into LDS: a staging thread loads a
// 4 (rows) x 4 (columns) block with four coalesced float4 loads and writes, per column, the four rows as ONE 8-byte group of
// bf16 (hi tile and lo tile).  LDS image per operand and half: [row block of 4][column][4 x bf16]: a thread's four columns are 32
// contiguous bytes (two b128 writes, conflict-free across the wave), and a lane's operand for a 32-row stage is the two row
// blocks (2 kg, 2 kg + 1) of its column: two b64 reads, consecutive lanes on consecutive 8-byte slots (conflict-free).
// (operand rows fetched PT = 2 stages ahead through a register ring, k loop unrolled by 2 -- what fits under 128 VGPRs; see k_gemm_nt_b3)
template <bool FAST>
__device__ __forceinline__ void tn_b3_tile(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                           float* __restrict__ C, int ldc, int M, int N, int m0, int n0, int r_begin, int r_end,
                                           float* __restrict__ colsum, int colsum_rows, bool sum_tile) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HALF = (BK / 4) * BM * 4;            // bf16 elements of one [row block][column][4] tile (BM == BN)
    __bf16* Ts = (__bf16*)smem;                          // [stage][A | B][hi | lo][HALF]
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int wm = (wave >> 2) * 64, wn = (wave & 3) * 32;   // 2 x 4 waves, each 64 (m) x 32 (n)
    const int li = lane & 15, kg = lane >> 4;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    // staging: threads 0..255 take A, 256..511 take B; each a 4-row x 4-column block of the 32 x 128 stage tile
    const bool is_b = t >= 256;
    const int tt = t & 255, mg = tt & 31, rb = tt >> 5;    // columns 4 mg .. 4 mg + 3, rows 4 rb .. 4 rb + 3 of the stage
    const float* src = is_b ? B : A;
    const int ld = is_b ? ldb : lda, c0 = (is_b ? n0 : m0) + 4 * mg, cmax = is_b ? N : M;
    const bool al = (ld & 3) == 0 && ((size_t)src & 15) == 0;
    const bool do_sum = colsum != nullptr && sum_tile && !is_b;
    constexpr int PT = 2;
    f32x4 rr_[PT][4], csum = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](f32x4 (&rr)[4], int r0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = r0 + 4 * rb + e;
            if constexpr (FAST) rr[e] = r < r_end ? *(const f32x4*)(src + (size_t)r * ld + c0) : (f32x4){0, 0, 0, 0};
            else rr[e] = r < r_end ? ld4(src + (size_t)r * ld + c0, al && c0 + 3 < cmax, cmax - c0) : (f32x4){0, 0, 0, 0};
        }
    };
    // (the column sums are taken at the LDS store, where the rows are consumed anyway -- not at the load, which would wait for it)
    auto sstore = [&](int buf, const f32x4 (&rr)[4], int r0) {
        if (do_sum)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (r0 + 4 * rb + e < colsum_rows) csum += rr[e];
        __bf16* tile = Ts + ((buf * 2 + (is_b ? 1 : 0)) * 2) * HALF + (rb * BM + 4 * mg) * 4;
        bf16x8 h[2], l[2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {                   // column c of the block: its four rows
            const f32x4 v = {rr[0][c], rr[1][c], rr[2][c], rr[3][c]};
            const bf16x4 hi = __builtin_convertvector(v, bf16x4);
            const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { h[c >> 1][(c & 1) * 4 + e] = hi[e]; l[c >> 1][(c & 1) * 4 + e] = lo[e]; }
        }
        *(bf16x8*)(tile) = h[0];
        *(bf16x8*)(tile + 8) = h[1];
        *(bf16x8*)(tile + HALF) = l[0];
        *(bf16x8*)(tile + HALF + 8) = l[1];
    };
    const int nk = (r_end - r_begin + BK - 1) / BK;
#pragma unroll
    for (int s_ = 0; s_ < PT; ++s_)
        if (s_ < nk) gload(rr_[s_], r_begin + s_ * BK);
    if (nk > 0) sstore(0, rr_[0], r_begin);
    __syncthreads();
    // lane (li, kg): k slots = the 8 rows of row blocks 2 kg and 2 kg + 1, for its column
    auto frag = [&](const __bf16* tile, int col) {
        const bf16x4 p = *(const bf16x4*)(tile + ((2 * kg) * BM + col) * 4);
        const bf16x4 q = *(const bf16x4*)(tile + ((2 * kg + 1) * BM + col) * 4);
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = p[e]; f[4 + e] = q[e]; }
        return f;
    };
    for (int kt0 = 0; kt0 < nk; kt0 += PT) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int kt = kt0 + u, buf = kt & 1;
            if (kt < nk) {
                if (kt + PT < nk) gload(rr_[u], r_begin + (kt + PT) * BK);   // slot u held stage kt: in LDS since the last iteration
                const __bf16* ta = Ts + (buf * 2 + 0) * 2 * HALF;
                const __bf16* tb = Ts + (buf * 2 + 1) * 2 * HALF;
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) { bh[j] = frag(tb, wn + 16 * j + li); bl[j] = frag(tb + HALF, wn + 16 * j + li); }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 ah = frag(ta, wm + 16 * i + li), al_ = frag(ta + HALF, wm + 16 * i + li);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al_, bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                    }
                }
                if (kt + 1 < nk) sstore(buf ^ 1, rr_[(u + 1) % PT], r_begin + (kt + 1) * BK);
                __syncthreads();
            }
        }
    }
    // D: col = lane&15 (n), row = 4*(lane>>4)+reg (m)
    const int cn = lane & 15, cr = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn + 16 * j + cn;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + 16 * i + cr + r;
#ifdef MP_EXP_TN_NOATOMIC   // ablation (wrong sums): what do the fp32 atomics of the split-K epilogue cost?
                if (m < M) C[(size_t)m * ldc + n] = acc[i][j][r];
#else
                if (m < M) atomicAdd(C + (size_t)m * ldc + n, acc[i][j][r]);
#endif
            }
        }
    if (colsum != nullptr && sum_tile) {   // A-staging threads with the same mg hold partial sums of the same 4 columns
        __syncthreads();
        float* red = smem;
        for (int i = t; i < BM; i += NT_THREADS) red[i] = 0.f;
        __syncthreads();
        if (!is_b)
            for (int e = 0; e < 4; ++e) atomicAdd(&red[4 * mg + e], csum[e]);
        __syncthreads();
        for (int i = t; i < BM; i += NT_THREADS)
            if (m0 + i < M && red[i] != 0.f) atomicAdd(colsum + m0 + i, red[i]);
    }
}

template <bool FAST>
__global__ __launch_bounds__(NT_THREADS, 2) void k_gemm_tn_b3(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                    float* __restrict__ C, int ldc, int M, int N, int K, int rows_per_block,
                                                    float* __restrict__ colsum, int colsum_rows) {
    const int r_begin = blockIdx.z * rows_per_block;
    tn_b3_tile<FAST>(A, lda, B, ldb, C, ldc, M, N, blockIdx.x * BM, blockIdx.y * BN, r_begin, min(K, r_begin + rows_per_block), colsum,
                     colsum_rows, blockIdx.y == 0);
}

// GROUPED form: up to MP_TN_MAX_GROUPS independent contractions (the weight gradients of all layers of a network) in ONE grid.
// Launched one contraction at a time, each 256 x 256 output is split into ~128 row slices to fill the chip, and every slice adds
// its 128 x 128 partial with fp32 atomics: 8.4 M atomics per layer -- about a third of the kernel's time.  With G layers in one
// grid the same number of workgroups needs 1/G of the slices per output (and the slices are long enough to amortise a
// workgroup's prologue / epilogue).  Aligned operands only (FAST path: 16 B aligned, M, N multiples of 128).
struct TnGroups {
    MpTnGroup g[MP_TN_MAX_GROUPS];
    int slice0[MP_TN_MAX_GROUPS + 1];    // first blockIdx.z of every group
    int n, rows_per_block;
};
// Workgroup -> (row slice, tile): the mt x nt tiles of ONE row slice read the same rows of A and B (each column block twice), so
// they are placed on the SAME XCD next to each other in dispatch order -- workgroup L runs on XCD L % 8 (MI355X_MICROARCH.md),
// hence tile = (L / 8) % tiles, slice = L % 8 + 8 (L / (8 tiles)) -- and the second reader finds the rows in that XCD's L2.
// With the (tile fastest, slice slowest) order of a plain 3-D grid the four tiles of a slice sit on four different XCDs and every
// operand byte crosses the fabric twice: the kernel then runs at the memory system's rate on 2x its algorithmic bytes.
__global__ __launch_bounds__(NT_THREADS, 2) void k_gemm_tn_b3g(TnGroups T, int mt, int nt, int n_slice) {
    const int tiles = mt * nt, L = blockIdx.x;
    const int tile = (L >> 3) % tiles, slice = (L & 7) + 8 * (L / (8 * tiles));
    if (slice >= n_slice) return;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MP_TN_MAX_GROUPS; ++i)
        if (i < T.n && slice >= T.slice0[i]) gi = i;
    const MpTnGroup& G = T.g[gi];
    const int bx = tile % mt, by = tile / mt;
    const int m0 = bx * BM, n0 = by * BN;
    if (m0 >= G.M || n0 >= G.N) return;
    const int r_begin = (slice - T.slice0[gi]) * T.rows_per_block;
    tn_b3_tile<true>(G.A, G.lda, G.B, G.ldb, G.C, G.ldc, G.M, G.N, m0, n0, r_begin, min(G.K, r_begin + T.rows_per_block), G.colsum,
                     G.colsum_rows, by == 0);
}


// WIDE form of the grouped contraction: ONE workgroup computes the whole 256 x 256 output of its row slice, so every operand row
// is read from HBM exactly once.  [The 128 x 128 tiles above read every column block twice; their four workgroups of a slice
// sit on one XCD but drift apart by more than its 4 MB L2 holds, the second read misses, and the kernel runs at the HBM rate on
// 2x its algorithmic bytes: 10.6 k cycles per 32-row stage, the matrix pipe 15 % busy.]  One workgroup per CU, 8 waves as 2 x 4,
// each a 128 (m) x 64 (n) sub-tile = 128 accumulator registers per lane; a 32-row stage is 64 KB of operands, staged as above
// (register transposition to [row block of 4][column][4 x bf16], hi and lo tiles), ONE stage ahead in registers (a second
// stage in flight spills: 1.2 ms).  Measured (15 contractions of 50-60 k rows, one body): 0.65 ms = 3 TB/s, against 0.99 ms of
// the 128 x 128 form.  Ablations: without the global loads 0.52 ms, without the matrix instructions 0.53 ms, plain stores
// instead of atomics 0.62 ms: a stage costs ~10 k cycles of LDS staging + fragment reads + conversions + MFMA that the
// lock-step of one barrier per stage does not overlap, and the loads fly only during the compute phase of one stage.
constexpr int WT_ = 256;                                  // output tile (both ways)
constexpr int WT_RB = WT_;                                // 8-byte slots between row blocks (padding them apart by 8 changed nothing)
constexpr int LDS_B3W = 2 * 2 * 2 * (BK / 4) * WT_RB * 4 * 2;    // [stage][A | B][hi | lo][row block][column][4] bf16 = 128 KB
#ifndef MP_EXP_TNW
#define MP_EXP_TNW 0
#endif
__global__ __launch_bounds__(NT_THREADS, 1) void k_gemm_tn_b3w(TnGroups T, int n_slice) {
    const int slice = blockIdx.x;
    if (slice >= n_slice) return;
    int gi = 0;
#pragma unroll
    for (int i = 1; i < MP_TN_MAX_GROUPS; ++i)
        if (i < T.n && slice >= T.slice0[i]) gi = i;
    const MpTnGroup& G = T.g[gi];
    const int r_begin = (slice - T.slice0[gi]) * T.rows_per_block, r_end = min(G.K, r_begin + T.rows_per_block);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HALF = (BK / 4) * WT_RB * 4;              // bf16 elements of one [row block][column][4] tile
    __bf16* Ts = (__bf16*)smem;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int li = lane & 15, kg = lane >> 4;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
    // staging: threads 0..255 take A, 256..511 take B; each TWO 4-row x 4-column blocks (row blocks rb and rb + 4) of the stage
    const bool is_b = t >= 256;
    const int tt = t & 255, mg = tt & 63, rb = tt >> 6;
    const float* src = (is_b ? G.B : G.A) + 4 * mg;
    const int ld = is_b ? G.ldb : G.lda;
    float* colsum = G.colsum;
    const int colsum_rows = G.colsum_rows;
    const bool do_sum = colsum != nullptr && !is_b;
#ifndef MP_TNW_PT
#define MP_TNW_PT 1
#endif
    constexpr int PT = MP_TNW_PT;      // stages in flight in registers (32 registers each)
    f32x4 rr_[PT][8], csum = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](f32x4 (&rr)[8], int r0) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + 4 * (rb + 4 * h) + e;
#if MP_EXP_TNW & 2      // ablation: no global loads
                rr[4 * h + e] = (f32x4){(float)r, 1.f, 2.f, 3.f};
#else
                rr[4 * h + e] = r < r_end ? *(const f32x4*)(src + (size_t)r * ld) : (f32x4){0, 0, 0, 0};
#endif
            }
    };
    // (the bias gradient's column sums are taken HERE, where the rows are consumed anyway: summed in gload they made the staging
    // waves wait for their loads at once -- every stage paid the full memory latency)
    auto sstore = [&](int buf, const f32x4 (&rr)[8], int r0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (do_sum)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (r0 + 4 * (rb + 4 * h) + e < colsum_rows) csum += rr[4 * h + e];
            __bf16* tile = Ts + ((buf * 2 + (is_b ? 1 : 0)) * 2) * HALF + ((rb + 4 * h) * WT_RB + 4 * mg) * 4;
            bf16x8 hh[2], ll[2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 v = {rr[4 * h][c], rr[4 * h + 1][c], rr[4 * h + 2][c], rr[4 * h + 3][c]};
                const bf16x4 hi = __builtin_convertvector(v, bf16x4);
                const bf16x4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4), bf16x4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { hh[c >> 1][(c & 1) * 4 + e] = hi[e]; ll[c >> 1][(c & 1) * 4 + e] = lo[e]; }
            }
            *(bf16x8*)(tile) = hh[0];
            *(bf16x8*)(tile + 8) = hh[1];
            *(bf16x8*)(tile + HALF) = ll[0];
            *(bf16x8*)(tile + HALF + 8) = ll[1];
        }
    };
    const int nk = (r_end - r_begin + BK - 1) / BK;
#pragma unroll
    for (int s_ = 0; s_ < PT; ++s_)
        if (s_ < nk) gload(rr_[s_], r_begin + s_ * BK);
    if (nk > 0) sstore(0, rr_[0], r_begin);
    __syncthreads();
    auto frag = [&](const __bf16* tile, int col) {
        const bf16x4 p = *(const bf16x4*)(tile + ((2 * kg) * WT_RB + col) * 4);
        const bf16x4 q = *(const bf16x4*)(tile + ((2 * kg + 1) * WT_RB + col) * 4);
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = p[e]; f[4 + e] = q[e]; }
        return f;
    };
    for (int kt0 = 0; kt0 < nk; kt0 += PT) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int kt = kt0 + u, buf = kt & 1;
            if (kt < nk) {
                if (kt + PT < nk) gload(rr_[u], r_begin + (kt + PT) * BK);
                const __bf16* ta = Ts + (buf * 2 + 0) * 2 * HALF;
                const __bf16* tb = Ts + (buf * 2 + 1) * 2 * HALF;
                // the n half (32 columns) outermost: its four B fragments stay in registers, A's are read once per half (all 256
                // registers of a lane are taken: 128 accumulators, 64 of the two stages in flight)
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    bf16x8 bh[2], bl[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        bh[j] = frag(tb, wn + 32 * jh + 16 * j + li);
                        bl[j] = frag(tb + HALF, wn + 32 * jh + 16 * j + li);
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const bf16x8 ah = frag(ta, wm + 16 * i + li), al_ = frag(ta + HALF, wm + 16 * i + li);
#if MP_EXP_TNW & 1      // ablation: no matrix instructions
                        acc[i][2 * jh][0] += (float)ah[0] + (float)al_[0] + (float)bh[0][0] + (float)bl[0][0] + (float)bh[1][0] + (float)bl[1][0];
                        continue;
#endif
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][2 * jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al_, bh[j], acc[i][2 * jh + j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][2 * jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][2 * jh + j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][2 * jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][2 * jh + j], 0, 0, 0);
                        if (i & 1) __builtin_amdgcn_sched_barrier(0);      // (keeps the scheduler from hoisting every fragment read)
                    }
                }
                if (kt + 1 < nk) sstore(buf ^ 1, rr_[(u + 1) % PT], r_begin + (kt + 1) * BK);
                __syncthreads();
            }
        }
    }
    const int cn = lane & 15, cr = (lane >> 4) * 4;
    float* Cm = G.C;
    const int ldc = G.ldc;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = wn + 16 * j + cn;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#if MP_EXP_TNW & 4      // ablation: plain stores instead of the split-K atomics (wrong sums)
                Cm[(size_t)(wm + 16 * i + cr + r) * ldc + n] = acc[i][j][r];
#else
                atomicAdd(Cm + (size_t)(wm + 16 * i + cr + r) * ldc + n, acc[i][j][r]);
#endif
        }
    if (colsum != nullptr) {   // A-staging threads with the same mg hold partial sums of the same 4 columns
        __syncthreads();
        float* red = smem;
        for (int i = t; i < WT_; i += NT_THREADS) red[i] = 0.f;
        __syncthreads();
        if (!is_b)
            for (int e = 0; e < 4; ++e) atomicAdd(&red[4 * mg + e], csum[e]);
        __syncthreads();
        for (int i = t; i < WT_; i += NT_THREADS)
            if (red[i] != 0.f) atomicAdd(colsum + i, red[i]);
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

#ifndef MP_EXP_NTP
#define MP_EXP_NTP 0
#endif
#ifndef MP_NT_PERSIST   // 1 = the 256-deep products of enough rows take the persistent weights-in-registers kernel (k_gemm_nt_b3p)
#define MP_NT_PERSIST 0
#endif
#ifndef MP_NT_WIDE      // ablation switch: 1 = layers wider than 128 take the 128 x 256 single-stage tile (activation rows read once;
#define MP_NT_WIDE 0    // 150 VGPRs, one workgroup per CU: 58 us per 50k x 256 x 256 product); 0 (default) = 128 x 128 two-stage: 49 us
#endif
extern "C" int mp_gemm_nt_bf16x3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                                 const float* bias, int bias_rows, int accumulate, int relu, void* stream) {
    if (M <= 0 || N <= 0) return 0;
    constexpr int LDS_22 = 2 * 2 * (BM + 128) * LDK16 * 2;   // 2 stages x (hi, lo) x (A tile + 128-row B tile) of bf16: 80 KB
    constexpr int LDS_41 = 1 * 2 * (BM + 256) * LDK16 * 2;   // 1 stage, 256-row B tile: 60 KB
    MP_LDS_ATTR((k_gemm_nt_b3<true, 1, 2, 2>), LDS_22);
    MP_LDS_ATTR((k_gemm_nt_b3<false, 1, 2, 2>), LDS_22);
    MP_LDS_ATTR((k_gemm_nt_b3<true, 1, 4, 1>), LDS_41);
    MP_LDS_ATTR((k_gemm_nt_b3<false, 1, 4, 1>), LDS_41);
    const bool fast = (lda & 3) == 0 && (ldb & 3) == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0 && K % BK == 0;
    // the persistent kernel: 256-deep products whose last 128-column block is at least half used, enough row tiles to go round
    const int row_tiles = (M + BM - 1) / BM, col_blocks = (N + 127) / 128;
    if (MP_NT_PERSIST && fast && !accumulate && K == NKP * BK && (N % 128 == 0 || N % 128 >= 64) && row_tiles * col_blocks >= 256) {
        constexpr int LDS_P = 2 * 2 * BM * LDK16 * 2;         // two stages x (hi, lo) x the 128-row activation tile of bf16: 40 KB
        MP_LDS_ATTR(k_gemm_nt_b3p, LDS_P);
        int gx = 256 / col_blocks;                            // one workgroup per CU
        if (gx > row_tiles) gx = row_tiles;
        if (gx < 1) gx = 1;
        hipLaunchKernelGGL(k_gemm_nt_b3p, dim3(gx, col_blocks), dim3(NT_THREADS), LDS_P, (hipStream_t)stream, A, lda, B, ldb, C, ldc,
                           M, N, bias, bias_rows, relu, row_tiles);
        return (int)hipGetLastError();
    }
    const bool wide = MP_NT_WIDE && N > 128;
    const dim3 grid((M + BM - 1) / BM, wide ? (N + 255) / 256 : (N + 127) / 128);
#define MP_NT_B3(F, J, S, L) hipLaunchKernelGGL((k_gemm_nt_b3<F, 1, J, S>), grid, dim3(NT_THREADS), L, (hipStream_t)stream, A, lda, B, ldb, \
                                                 C, ldc, M, N, K, bias, bias_rows, accumulate, relu)
    if (wide) { if (fast) MP_NT_B3(true, 4, 1, LDS_41); else MP_NT_B3(false, 4, 1, LDS_41); }
    else      { if (fast) MP_NT_B3(true, 2, 2, LDS_22); else MP_NT_B3(false, 2, 2, LDS_22); }
#undef MP_NT_B3
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

extern "C" int mp_gemm_tn_bf16x3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                                 float* colsum, int colsum_rows, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
#ifndef MP_TN_WGS       // workgroups the contraction is split into (tiles x row slices): every slice adds its 128 x 128 partial to C with
#define MP_TN_WGS 512   // fp32 atomics, so fewer slices = fewer atomics but less of the chip busy
#endif
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int slices = (MP_TN_WGS + tiles - 1) / tiles;
    int rows = (K + slices - 1) / slices;
    rows = (rows + BK - 1) / BK * BK;
    if (rows < 4 * BK) rows = 4 * BK;
    slices = (K + rows - 1) / rows;
    constexpr int LDS_B3 = 2 * 2 * 2 * (BK / 4) * BM * 4 * 2;   // 2 stages x (A, B) x (hi, lo) x [BK/4][128][4] bf16
    MP_LDS_ATTR((k_gemm_tn_b3<true>), LDS_B3);
    MP_LDS_ATTR((k_gemm_tn_b3<false>), LDS_B3);
    const bool fast = (lda & 3) == 0 && (ldb & 3) == 0 && ((size_t)A & 15) == 0 && ((size_t)B & 15) == 0 && M % BM == 0 &&
                      N % BN == 0;
    if (fast)
        hipLaunchKernelGGL(k_gemm_tn_b3<true>, dim3(M / BM, N / BN, slices), dim3(NT_THREADS), LDS_B3, (hipStream_t)stream, A, lda, B, ldb,
                           C, ldc, M, N, K, rows, colsum, colsum_rows);
    else
        hipLaunchKernelGGL(k_gemm_tn_b3<false>, dim3((M + BM - 1) / BM, (N + BN - 1) / BN, slices), dim3(NT_THREADS), LDS_B3,
                           (hipStream_t)stream, A, lda, B, ldb, C, ldc, M, N, K, rows, colsum, colsum_rows);
    return (int)hipGetLastError();
}

extern "C" int mp_gemm_tn_bf16x3_grouped(const MpTnGroup* groups, int n_groups, void* stream) {
    if (n_groups <= 0) return 0;
    if (n_groups > MP_TN_MAX_GROUPS) return -1;
    TnGroups T;
    T.n = n_groups;
    long long work = 0;
    int mt = 0, nt = 0;
    for (int i = 0; i < n_groups; ++i) {
        const MpTnGroup& g = groups[i];
        if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % BM || g.N % BN || (g.lda & 3) || (g.ldb & 3) || ((size_t)g.A & 15) ||
            ((size_t)g.B & 15))
            return -2;
        T.g[i] = g;
        work += (long long)(g.M / BM) * (g.N / BN) * g.K;
        mt = g.M / BM > mt ? g.M / BM : mt;
        nt = g.N / BN > nt ? g.N / BN : nt;
    }
#ifndef MP_TNW_WGS
#define MP_TNW_WGS 256    // workgroups of a wide launch: one per CU
#endif
    bool wide = true;
    for (int i = 0; i < n_groups; ++i) wide = wide && groups[i].M == WT_ && groups[i].N == WT_;
    if (wide) {
        long long rows = (work / 4 + MP_TNW_WGS - 1) / MP_TNW_WGS;
        rows = (rows + BK - 1) / BK * BK;
        if (rows < 8 * BK) rows = 8 * BK;
        int z = MP_TNW_WGS + 1;
        while (z > MP_TNW_WGS) {         // every group rounds its slice count up: lengthen the slices until ONE round holds them all
            z = 0;
            for (int i = 0; i < n_groups; ++i) z += (int)((groups[i].K + rows - 1) / rows);
            if (z > MP_TNW_WGS) rows += BK;
        }
        T.rows_per_block = (int)rows;
        z = 0;
        for (int i = 0; i < n_groups; ++i) {
            T.slice0[i] = z;
            z += (int)((groups[i].K + rows - 1) / rows);
        }
        T.slice0[n_groups] = z;
        MP_LDS_ATTR(k_gemm_tn_b3w, LDS_B3W);
        hipLaunchKernelGGL(k_gemm_tn_b3w, dim3(z), dim3(NT_THREADS), LDS_B3W, (hipStream_t)stream, T, z);
        return (int)hipGetLastError();
    }
#ifndef MP_TNG_WGS
#define MP_TNG_WGS 1024   // workgroups of a grouped launch: 2 resident per CU, two rounds
#endif
    long long rows = (work + MP_TNG_WGS - 1) / MP_TNG_WGS;
    rows = (rows + BK - 1) / BK * BK;
    if (rows < 8 * BK) rows = 8 * BK;
    for (;;) {                           // (as above: no third, nearly empty round of workgroups)
        long long wgs = 0;
        for (int i = 0; i < n_groups; ++i) wgs += (long long)(groups[i].M / BM) * (groups[i].N / BN) * ((groups[i].K + rows - 1) / rows);
        if (wgs <= MP_TNG_WGS) break;
        rows += BK;
    }
    T.rows_per_block = (int)rows;
    int z = 0;
    for (int i = 0; i < n_groups; ++i) {
        T.slice0[i] = z;
        z += (int)((groups[i].K + rows - 1) / rows);
    }
    T.slice0[n_groups] = z;
    constexpr int LDS_B3 = 2 * 2 * 2 * (BK / 4) * BM * 4 * 2;
    MP_LDS_ATTR(k_gemm_tn_b3g, LDS_B3);
    const int zpad = (z + 7) / 8 * 8;
    hipLaunchKernelGGL(k_gemm_tn_b3g, dim3(mt * nt * zpad), dim3(NT_THREADS), LDS_B3, (hipStream_t)stream, T, mt, nt, z);
    return (int)hipGetLastError();
}
