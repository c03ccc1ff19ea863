// Layer-FUSED training kernels of the foreground ImplicitNet (the reverse-over-reverse formulation of multiply_amd/train.py
// ImplicitTrainRev; reference: code/lib/model/networks.py:160-181 evaluated under torch autograd with create_graph=True,
// code/lib/model/multiply.py:620-661, optimiser step code/multiply_model.py:212-222).
//
// The unfused path runs one GEMM + separate element-wise passes per layer, every activation matrix round-tripping HBM in fp32
// (6 products + 5 passes per layer and person).  Here a workgroup keeps a tile of 128 points ON CHIP across the whole layer
// chain, exactly like the inference kernels of mlp_core.hpp do: the activations of a wave's 16 points are the MFMA's B operand
// and never leave the register file (the next layer's operand registers are written by the activation code of the previous one,
// through the same K-slot permutation), the weights stream global -> LDS by DMA (global_load_lds, 3-slot ring of 32-row chunks,
// shared by the 8 waves).  Arithmetic: the split-bfloat16 product of gemm.hip (x = hi + lo, three v_mfma_f32_16x16x32_bf16 per
// block: lo.hi + hi.lo + hi.hi, fp32 accumulation, ~2^-16 per product, the range of fp32); activations are computed in fp32.
//
//   k_tf_sdf_fwd : value sweep  Z_l = W_l X_l + b_l, X_{l+1} = softplus(Z_l)            (l = 0..8, skip connection at 4)
//                  then the gradient sweep  V_7 = s_7 (.) w8,  T_l = W_l^T V_l,  V_{l-1} = s_{l-1} (.) T_l   (l = 7..1),
//                  G = W_0[:, :39]^T V_0 + (the Fourier rows of T_4)      (d sdf / d PE; J_PE^T is applied by mp_tr_pe_grad_fwd)
//   k_tf_sdf_bwd : the adjoint of both sweeps w.r.t. the activations (the data path):
//                  ascending   dV_0 = W_0[:, :39] dG,  (dU_l, dS_l) = adj(s_l, U_l, dV_l),  dV_{l+1} = W_{l+1} dT_{l+1}
//                  descending  dX_l = W_l^T dZ_l,  dZ_{l-1} = s_{l-1} (.) dX_l + dS_{l-1}
// What leaves the chip is what the WEIGHT-gradient contractions need (X_l, V_l from the forward, dZ_l, dT_l from the backward:
// row-major fp32 [points][256], consumed by mp_gemm_tn_bf16x3 with K = 2 P: [dZ_l; V_l]^T [X_l; dT_l]) and what the adjoint
// needs (U_l = the gradient sweep's pre-sigmoid rows; dS_l between the two backward sweeps).  sigma' and sigma'' are re-derived
// from the stored X_{l+1} = softplus(Z_l):  1 - sigma' = exp(-100 X),  sigma'' = 100 sigma' (1 - sigma').
//
// Specialised for the shipped foreground network (confs/model/*.yaml: 8 x 256, skip_in [4], multires 6, d_in 3, 257 outputs):
// multiply_amd/train.py falls back to the layer-wise HIP path (ImplicitTrainRev) for any other shape.
// Entry points: include/multiply_hip.h (mp_tf_*).
#include <hip/hip_runtime.h>
#include "../../include/multiply_hip.h"
#include "common.hpp"
#include "mlp_core.hpp"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TF_WAVES = 8, TF_THREADS = 64 * TF_WAVES, TF_PTS = 16 * TF_WAVES;   // 128 points per workgroup
constexpr int TILE_B = 1024;                  // one 16 x 32 bf16 A tile in fragment order [lane][8]
constexpr int CH_REG = 2 * 8 * 2 * TILE_B;    // register-fed part of a chunk: [row block][K step][hi | lo] = 32 KiB
constexpr int CH_IN = 2 * 2 * 2 * TILE_B;     // input-fed part (Fourier features, 64 slots): [row block][K step][hi | lo] = 8 KiB
constexpr int CH_BYTES = CH_REG + CH_IN;      // 40 KiB
constexpr int RING = 3;
constexpr int BIN_BYTES = 4096;               // per wave: the input-fed B fragments [K step][hi | lo][lane][8]
constexpr int LDS_BYTES = RING * CH_BYTES + TF_WAVES * BIN_BYTES + 272 * 4;
constexpr float K2 = 144.26950408889634f;     // 100 log2(e): softplus_100 in base 2
constexpr float R2 = 0.70710678118654752f;    // the skip connection's 1/sqrt(2)
constexpr int E_PE = 39, OUT3 = 217, BIAS_LD = 288, HID = 256;

// ---- chunk stream layout (40 KiB chunks of 32 output rows) --------------------------------------------------------------------
//   0 .. 72   value orientation W_l, l = 0..8 (8 chunks each; layer 8 = 256 feature rows, then the sdf row: 9 chunks)
//  73 .. 130  transposed W_l^T, l = 7..1 (8 chunks each; l = 4: 10 chunks, the last two = its 39 Fourier rows)
// 131 .. 132  W_0[:, :39]^T (39 rows)
// 133 .. 196  a second run for the backward's descending sweep: W_8[1:]^T, then W_l^T for l = 7..1, 8 chunks each
// The forward consumes 0 .. 132 in order; the backward 0 .. 63, then 133 .. 196.
constexpr int CH_WT7 = 73, CH_WT4 = 97, CH_WT3 = 107, CH_G0 = 131, CH_WT8B = 133, CH_WTB = 141, CH_TOTAL = 197;
constexpr int FWD_CHUNKS = 133, BWD_CHUNKS = 128, BWD_SHIFT = CH_WT8B - 64;

__device__ __forceinline__ int slot_feature(int ks, int g, int e) { return 32 * ks + (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4); }

// value of pack element: chunk, local row r (0..31), register-fed K feature f (0..255) or (f < 0) input slot s (0..63)
__device__ float pack_value(const float* const* __restrict__ W, int chunk, int r, int f, int s) {
    const bool reg = f >= 0;
    if (chunk <= 72) {                                   // value orientation
        const int l = chunk == 72 ? 8 : chunk >> 3, R = chunk == 72 ? 256 + r : 32 * (chunk & 7) + r;
        int o;
        if (l < 8) { o = R; if (R >= (l == 3 ? OUT3 : HID)) return 0.f; }
        else { if (R > 256) return 0.f; o = R < 256 ? R + 1 : 0; }
        const int in_dim = l == 0 ? 108 : HID;
        if (reg) {
            if (l == 0) return 0.f;
            if (l == 4 && f >= OUT3) return 0.f;
            return W[l][(size_t)o * in_dim + f];
        }
        if (s >= E_PE) return 0.f;
        if (l == 0) return W[0][(size_t)o * in_dim + s];
        if (l == 4) return W[4][(size_t)o * in_dim + OUT3 + s] * R2;
        return 0.f;
    }
    if (!reg) return 0.f;
    int l, c;
    if (chunk < CH_WT4) { l = 7 - (chunk - CH_WT7) / 8; c = (chunk - CH_WT7) % 8; }
    else if (chunk < CH_WT3) { l = 4; c = chunk - CH_WT4; }
    else if (chunk < CH_G0) { l = 3 - (chunk - CH_WT3) / 8; c = (chunk - CH_WT3) % 8; }
    else if (chunk < CH_WT8B) { l = 0; c = chunk - CH_G0; }
    else if (chunk < CH_WTB) { l = 8; c = chunk - CH_WT8B; }
    else { l = 7 - (chunk - CH_WTB) / 8; c = (chunk - CH_WTB) % 8; }
    const int R = 32 * c + r;
    if (l == 0) return R < E_PE ? W[0][(size_t)f * 108 + R] : 0.f;            // W_0[:, :39]^T
    if (l == 8) return W[8][(size_t)(f + 1) * HID + R];                        // W_8[1:]^T
    if (f >= (l == 3 ? OUT3 : HID)) return 0.f;                                // K = the layer's outputs
    if (l == 4) {
        int i;
        if (R < HID) { if (R >= OUT3) return 0.f; i = R; }
        else { if (R - HID >= E_PE) return 0.f; i = OUT3 + R - HID; }
        return W[4][(size_t)f * HID + i] * R2;
    }
    return W[l][(size_t)f * HID + R];
}

template <class F>
__device__ __forceinline__ void pack_chunk(const float* const* __restrict__ W, __bf16* __restrict__ wpack, F value) {
    const int chunk = blockIdx.x;
    __bf16* dst = wpack + (size_t)chunk * (CH_BYTES / 2);
    for (int idx = threadIdx.x; idx < 2 * 10 * 64 * 8; idx += 512) {
        const int e = idx & 7, lane = (idx >> 3) & 63, kk = (idx >> 9) % 10, mb = idx / 5120;
        const int r = 16 * mb + (lane & 15), g = lane >> 4;
        float v;
        size_t o;
        if (kk < 8) {
            v = value(W, chunk, r, slot_feature(kk, g, e), -1);
            o = (size_t)((mb * 8 + kk) * 2) * 512 + lane * 8 + e;
        } else {
            v = value(W, chunk, r, -1, 32 * (kk - 8) + 8 * g + e);
            o = (size_t)(CH_REG / 2) + (size_t)((mb * 2 + (kk - 8)) * 2) * 512 + lane * 8 + e;
        }
        const __bf16 hi = (__bf16)v;
        dst[o] = hi;
        dst[o + 512] = (__bf16)(v - (float)hi);
    }
}

__global__ __launch_bounds__(512) void k_tf_pack(const float* const* __restrict__ W, const float* const* __restrict__ B,
                                                 __bf16* __restrict__ wpack, float* __restrict__ bias_all) {
    pack_chunk(W, wpack, pack_value);
    const int chunk = blockIdx.x;
    if (chunk < 9) {                                      // the biases in pack-row order
        const int l = chunk;
        for (int r = threadIdx.x; r < BIAS_LD; r += 512) {
            float b = 0.f;
            if (l < 8) { if (r < (l == 3 ? OUT3 : HID)) b = B[l][r]; }
            else if (r <= 256) b = B[8][r < 256 ? r + 1 : 0];
            bias_all[l * BIAS_LD + r] = b;
        }
    }
}

// ---- stash arena (floats), P points.  Every [P][256] tensor has P + 1 rows: row P is where the lanes of the last tile's missing
// points store (unconditional stores, no select: a divergent branch around them makes hipcc's wait-count bookkeeping fall back
// to vmcnt(0) at every chunk).  R1 = (P + 1) * 256:
//   dZ(l) l = 0..7 at l R1 (backward)          V(l)  l = 0..7 at (8 + l) R1 (forward)
//   X(l)  l = 1..8 at (15 + l) R1 (forward)    dT(l) l = 1..7 at (23 + l) R1 (backward)
//   U(l)  l = 0..6 at (31 + l) R1: the gradient sweep's rows before the sigmoid factor
//   dS(l) l = 0..7 at (38 + l) R1: sigma'' (.) U (.) dV between the backward's two sweeps
//   IN [P][39] at 46 R1 (Fourier features), dG [P][39] at 46 R1 + 39 P, G [P][39] at 46 R1 + 78 P
struct TfArgs {
    const char* wpack;
    const float* bias;     // [9][288]
    const float* w8;       // [256]: the sdf row of the last layer
    float* arena;
    float* feat;           // fwd: [P][256] feature rows of the last layer
    float* sdf;            // fwd: [P]
    const float* dfeat;    // bwd: [P][256]
    const float* dsdf;     // bwd: [P]
    float* dw8;            // bwd: [256] +=  gradient of the last layer's sdf row
    float* db8;            // bwd: [1] +=  gradient of its bias
    int P;
};
__host__ __device__ inline size_t off_dZ(size_t R1, int l) { return (size_t)l * R1; }
__host__ __device__ inline size_t off_V(size_t R1, int l) { return (size_t)(8 + l) * R1; }
__host__ __device__ inline size_t off_X(size_t R1, int l) { return (size_t)(15 + l) * R1; }
__host__ __device__ inline size_t off_dT(size_t R1, int l) { return (size_t)(23 + l) * R1; }
__host__ __device__ inline size_t off_U(size_t R1, int l) { return (size_t)(31 + l) * R1; }
__host__ __device__ inline size_t off_dS(size_t R1, int l) { return (size_t)(38 + l) * R1; }
__host__ __device__ inline size_t off_IN(size_t R1) { return 46 * R1; }
__host__ __device__ inline size_t off_dG(size_t R1, size_t P) { return 46 * R1 + 39 * P; }
__host__ __device__ inline size_t off_G(size_t R1, size_t P) { return 46 * R1 + 78 * P; }

struct BReg { bf16x8 h[8], l[8]; };
__device__ __forceinline__ bf16x8 zero_frag() { return __builtin_bit_cast(bf16x8, (f32x4){0.f, 0.f, 0.f, 0.f}); }

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    const bf16x4 ha = __builtin_convertvector(a, bf16x4), hb = __builtin_convertvector(b, bf16x4);
    const bf16x4 la = __builtin_convertvector(a - __builtin_convertvector(ha, f32x4), bf16x4);
    const bf16x4 lb = __builtin_convertvector(b - __builtin_convertvector(hb, f32x4), bf16x4);
    hi = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct Ctx {
    const char* wpack;
    char* ring;
    char* binf;        // this wave's input-fragment block
    float* redf;       // [257] per-workgroup column sums (backward)
    int wave, lane, g, j;
    bool late;         // second wave of its SIMD: barrier between the products and the activation code of a chunk
    int ci, ring_pos, n_total;
    int split_at, src_shift;            // stream position k -> chunk k (k < split_at) or k + src_shift
    int noreg_hi, in0_lo, in0_hi, in1_lo, in1_hi;   // chunks below noreg_hi have no register-fed part; two chunk ranges have an input-fed part
    bool valid;
    unsigned row;      // 256 * point index, clamped to the last point: LOADS of this lane's stash rows
    unsigned srow;     // 256 * min(point index, P): STORES (row P = the tensors' pad row)
    unsigned pad;      // P
    size_t prow;       // (clamped) point index
    float* trash;      // 256 floats for the few stores into caller-owned arrays without a pad row
};

// pieces [wl, wl + nw, ...) of chunk k into ring slot `slot`
template <int NW>
__device__ __forceinline__ void tf_issue(const Ctx& cx, int k, int slot, int wl) {
    const int ku = __builtin_amdgcn_readfirstlane(k), wu = __builtin_amdgcn_readfirstlane(wl);
    const int src = ku < cx.split_at ? ku : ku + cx.src_shift;
    const bool has_reg = src >= cx.noreg_hi, has_in = (src >= cx.in0_lo && src < cx.in0_hi) || (src >= cx.in1_lo && src < cx.in1_hi);
    const char* s = mp::uniform_ptr(cx.wpack) + (size_t)src * CH_BYTES + wu * TILE_B;
    const unsigned d = __builtin_amdgcn_readfirstlane(mp::lds_offset(cx.ring)) + __builtin_amdgcn_readfirstlane(slot) * CH_BYTES +
                       wu * TILE_B;
#if TF_EXP & 8
    return;
#endif
    if (has_reg) {
#pragma unroll
        for (int i = 0; i < 32 / NW; ++i) mp::lds_dma_16(s + i * NW * TILE_B, cx.lane * 16, d + i * NW * TILE_B);
    }
    if (has_in) {
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) mp::lds_dma_16(s + CH_REG + i * NW * TILE_B, cx.lane * 16, d + CH_REG + i * NW * TILE_B);
    }
}

// ablation switches (timing experiments only: results are wrong), -DTF_EXP=<bits>: 1 no stash stores, 2 no MFMAs, 4 no A-fragment
// LDS reads, 8 no weight DMA, 16 no stash loads (the activation code sees zeros)
#ifndef TF_EXP
#define TF_EXP 0
#endif
#if TF_EXP & 2
#define TF_MFMA(a, b, c) (c)
#else
#define TF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
#ifndef TF_QD
#define TF_QD 3      // depth of the A-fragment register queue (K steps)
#endif

__device__ __forceinline__ bf16x8 lds_frag(const char* p) {
#if TF_EXP & 4
    return __builtin_bit_cast(bf16x8, (f32x4){1e-3f, 2e-3f, 1e-3f, 2e-3f});
#else
    return *(const bf16x8*)p;
#endif
}
// the products of one chunk: acc[mb] (16 rows x 16 points) += W tile . activations, K = 256 from registers (+ 64 from the input)
template <bool HAS_IN>
__device__ __forceinline__ void tf_mma(const Ctx& cx, bool use_reg, bool use_in, const BReg& B, f32x4 (&acc)[2]) {
    const char* slot = cx.ring + cx.ring_pos * CH_BYTES + cx.lane * 16;
    if (use_reg) {
        // A fragments TF_QD - 1 K steps ahead in a register queue: q[.][2 mb + half].  Pinned with sched_barrier: left alone,
        // hipcc sinks every ds_read to just before its MFMA and drains lgkmcnt(0) sixteen times per chunk.  One K step ahead is
        // ~100 cycles of MFMA cover -- less than the LDS latency with eight waves reading 4 KB per K step each.
        bf16x8 q[TF_QD][4];
#pragma unroll
        for (int k0 = 0; k0 < TF_QD - 1; ++k0)
#pragma unroll
            for (int t = 0; t < 4; ++t) q[k0][t] = lds_frag(slot + (((t >> 1) * 8 + k0) * 2 + (t & 1)) * TILE_B);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + TF_QD - 1 < 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    q[(ks + TF_QD - 1) % TF_QD][t] = lds_frag(slot + (((t >> 1) * 8 + ks + TF_QD - 1) * 2 + (t & 1)) * TILE_B);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 (&a)[4] = q[ks % TF_QD];   // a[0] = hi of row block 0, a[1] = lo, a[2] = hi of row block 1, a[3] = lo
            // the small terms first; the two row blocks alternate so that no MFMA waits for the one before it
            acc[0] = TF_MFMA(a[1], B.h[ks], acc[0]);
            acc[1] = TF_MFMA(a[3], B.h[ks], acc[1]);
            acc[0] = TF_MFMA(a[0], B.l[ks], acc[0]);
            acc[1] = TF_MFMA(a[2], B.l[ks], acc[1]);
            acc[0] = TF_MFMA(a[0], B.h[ks], acc[0]);
            acc[1] = TF_MFMA(a[2], B.h[ks], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (HAS_IN) {
        if (use_in) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 bh = *(const bf16x8*)(cx.binf + (ks * 2 + 0) * TILE_B + cx.lane * 16);
                const bf16x8 bl = *(const bf16x8*)(cx.binf + (ks * 2 + 1) * TILE_B + cx.lane * 16);
                bf16x8 ah[2], al[2];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    ah[mb] = *(const bf16x8*)(slot + CH_REG + ((mb * 2 + ks) * 2 + 0) * TILE_B);
                    al[mb] = *(const bf16x8*)(slot + CH_REG + ((mb * 2 + ks) * 2 + 1) * TILE_B);
                }
                acc[0] = TF_MFMA(al[0], bh, acc[0]);
                acc[1] = TF_MFMA(al[1], bh, acc[1]);
                acc[0] = TF_MFMA(ah[0], bl, acc[0]);
                acc[1] = TF_MFMA(ah[1], bl, acc[1]);
                acc[0] = TF_MFMA(ah[0], bh, acc[0]);
                acc[1] = TF_MFMA(ah[1], bh, acc[1]);
            }
        }
    }
}

// One layer = up to MAXC chunks of 32 output rows.  Per chunk: the products, the chunk barrier, the activation code (Epi::run),
// which also writes K step c of the NEXT layer's operand.  Waves 0..3 ("early", one per SIMD) pass the barrier behind their
// activation code, waves 4..7 ("late") between products and activation code: the two waves of a SIMD run in anti-phase, one
// wave's VALU / memory work beside the other's MFMAs (mlp_core.hpp run_layer_pp).
// Ring protocol: behind barrier(ci) every wave is done with chunk ci's products, so the late waves refill its slot with chunk
// ci + 3 and make sure those pieces have landed before barrier(ci + 1); chunk ci + 3 is first read behind barrier(ci + 2).
// Memory latency: what the activation code of chunk c + 1 reads (biases, stash rows) is requested at the head of the activation
// code of chunk c, BEFORE chunk c's stash stores -- vector memory operations complete in issue order, so a load issued behind a
// store cannot return before the store is acknowledged (measured on the first version: ~2.5 us per chunk = one HBM round trip,
// against 0.8 us of MFMA work).  For the same reason nothing here waits with vmcnt(0): Epi::touch() makes the compiler wait for
// the prefetched registers at a point where only stores are younger (and before the asm-issued DMA, which it does not know of),
// and the explicit wait is counted: vmcnt(npref) = "everything older than the last npref operations", i.e. older than the
// prefetch loads, which are younger than the DMA pieces of the previous barrier.
template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }
__device__ __forceinline__ void wait_vm_rt(int n) {
    if (n >= 4) wait_vm<4>();
    else if (n >= 2) wait_vm<2>();
    else wait_vm<0>();
}
__device__ __forceinline__ void touch4(const f32x4& v) { asm volatile("" ::"v"(v)); }

// EARLY_DMA (round 6; the value-only kernel -- its activation code issues no stores; -DMP_DMA_LATE: the round-5 schedule): the weight DMA is issued by the EARLY waves
// at the tail of their activation code -- where they otherwise wait at the barrier for the late waves, whose [barrier, DMA issue,
// activation, products] chain is the chunk's critical path.  Behind barrier(ci - 1) slot (ci - 1) % 3 is free: chunk ci + 2 goes
// there, and the issuing wave waits for it one iteration later (counted: everything but the prefetch loads just issued).
// DMA_MODE 2 (-DMP_DMA_SPLIT): both halves issue, NW = 8 piece dealing -- the late waves their pieces of chunk ci + 3 behind the barrier,
// the early waves theirs of chunk ci + 2 at their tail.
template <class Epi, int MAXC, bool HAS_IN, int DMA_MODE = 0>
__device__ __forceinline__ void tf_layer(Ctx& cx, Epi& ep, int n_chunk, bool use_reg, bool use_in, BReg& Bcur, BReg& Bnext) {
    int npref = ep.prefetch(cx, 0);
    ep.rotate();
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < n_chunk) {
            f32x4 acc[2];
            ep.init(cx, c, acc);
            tf_mma<HAS_IN>(cx, use_reg, use_in, Bcur, acc);
            ep.touch();
            constexpr bool EARLY_DMA = DMA_MODE != 0;
            if (cx.late) {
                if constexpr (DMA_MODE != 1) wait_vm_rt(npref);
                __syncthreads();
                if constexpr (DMA_MODE == 0) {
                    if (cx.ci + RING < cx.n_total) tf_issue<4>(cx, cx.ci + RING, cx.ring_pos, cx.wave - 4);
                } else if constexpr (DMA_MODE == 2) {
                    if (cx.ci + RING < cx.n_total) tf_issue<8>(cx, cx.ci + RING, cx.ring_pos, cx.wave);
                }
            }
            npref = c + 1 < n_chunk ? ep.prefetch(cx, c + 1) : 0;
            ep.run(cx, c, acc, Bnext);
            ep.rotate();
            if constexpr (EARLY_DMA) {
                if (!cx.late) {
                    wait_vm_rt(npref);
                    if (cx.ci >= 1 && cx.ci + RING - 1 < cx.n_total) {
                        if constexpr (DMA_MODE == 2) tf_issue<8>(cx, cx.ci + RING - 1, cx.ring_pos == 0 ? RING - 1 : cx.ring_pos - 1, cx.wave);
                        else tf_issue<4>(cx, cx.ci + RING - 1, cx.ring_pos == 0 ? RING - 1 : cx.ring_pos - 1, cx.wave);
                    }
                }
            }
            if (!cx.late) __syncthreads();
            ++cx.ci;
            cx.ring_pos = cx.ring_pos + 1 == RING ? 0 : cx.ring_pos + 1;
        }
    }
    if constexpr (Epi::NEXT_FROM_MEM) {
        ep.load_next(cx, Bcur);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { Bcur.h[k] = Bnext.h[k]; Bcur.l[k] = Bnext.l[k]; }
    }
}

#if TF_EXP & 16
__device__ __forceinline__ f32x4 ld4g(const float* p) { return (f32x4){0.01f, 0.02f, 0.03f, 0.04f}; }
#else
__device__ __forceinline__ f32x4 ld4g(const float* p) { return *(const f32x4*)p; }
#endif
__device__ __forceinline__ void st4g(float* p, f32x4 v) {
#if TF_EXP & 1
    if (v[0] == 123.456f)
#endif
        *(f32x4*)p = v;
}
// 1 - sigma'(Z) = exp(-100 softplus(Z)) from the stored activation x = scale * softplus(Z), kx = 100 log2(e) / scale
// (clamped at x = 0: columns 217.. of X_4 hold the re-injected Fourier features, which may be negative -- those columns only ever
// meet zero weights, but 2^(+144) = inf times a zero weight would be a NaN inside the MFMA)
__device__ __forceinline__ float one_minus_sig(float x, float kx) { return __builtin_amdgcn_exp2f(__builtin_fminf(-(x * kx), 0.0f)); }

__device__ __forceinline__ void ctx_setup(Ctx& cx, char* smem, const char* wpack, int P, int n_total, int split_at, int src_shift,
                                          float* trash) {
    cx.trash = trash;
    cx.wpack = wpack;
    cx.ring = smem;
    cx.wave = threadIdx.x >> 6;
    cx.lane = threadIdx.x & 63;
    cx.g = cx.lane >> 4;
    cx.j = cx.lane & 15;
    cx.binf = smem + RING * CH_BYTES + cx.wave * BIN_BYTES;
    cx.redf = (float*)(smem + RING * CH_BYTES + TF_WAVES * BIN_BYTES);
    cx.late = __builtin_amdgcn_readfirstlane(cx.wave) >= TF_WAVES / 2;
    cx.ci = 0;
    cx.ring_pos = 0;
    cx.n_total = n_total;
    cx.split_at = split_at;
    cx.src_shift = src_shift;
    cx.noreg_hi = 8; cx.in0_lo = 0; cx.in0_hi = 8; cx.in1_lo = 32; cx.in1_hi = 40;     // the SDF net's stream (colour kernels override)
    const int pt = blockIdx.x * TF_PTS + cx.wave * 16 + cx.j;
    cx.valid = pt < P;
    cx.prow = (size_t)(pt < P ? pt : P - 1);
    cx.row = (unsigned)cx.prow * HID;
    cx.srow = (unsigned)(pt < P ? pt : P) * HID;
    cx.pad = (unsigned)P;
}

// the wave's input-fed B fragments (64 K slots in natural order, 39 used) of src [P][39] into its LDS block
__device__ __forceinline__ void build_bin(const Ctx& cx, const float* __restrict__ src, int n_in = E_PE) {
    const float* p = src + cx.prow * n_in;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int s = 32 * ks + 8 * cx.g + e;
            v[e] = s < n_in ? p[s] : 0.0f;
        }
        bf16x8 hi, lo;
        split8(v, hi, lo);
        *(bf16x8*)(cx.binf + (ks * 2 + 0) * TILE_B + cx.lane * 16) = hi;
        *(bf16x8*)(cx.binf + (ks * 2 + 1) * TILE_B + cx.lane * 16) = lo;
    }
}

__device__ __forceinline__ void tf_prologue(Ctx& cx) {
#pragma unroll
    for (int k = 0; k < RING; ++k) tf_issue<TF_WAVES>(cx, k, k, cx.wave);
    mp::dma_wait_all();
    __syncthreads();
}

// ================================================================================================================ forward
// value sweep: softplus layers (and the linear last layer)
struct EpiA {
    static constexpr bool NEXT_FROM_MEM = false;
    const float* bias;     // this layer's biases, pack-row order
    float* xout;           // X_{l+1} rows (row-major [P][256])
    float osc;             // scale / K2 of the stored activation (layer 3: 1/sqrt(2): the skip connection's factor)
    bool linear;           // layer 8: rows 0..255 -> feat [P][256], row 256 -> sdf [P]
    float* feat;
    float* sdf;
    f32x4 pb[2], nb[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        nb[0] = ld4g(bias + 32 * c + 4 * cx.g);
        nb[1] = ld4g(bias + 32 * c + 16 + 4 * cx.g);
        return 2;
    }
    __device__ __forceinline__ void rotate() { pb[0] = nb[0]; pb[1] = nb[1]; }
    __device__ __forceinline__ void touch() { touch4(pb[0]); touch4(pb[1]); }
    __device__ __forceinline__ void init(const Ctx&, int, f32x4 (&acc)[2]) {
        acc[0] = pb[0];
        acc[1] = pb[1];
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        if (linear) {
            if (c < 8) {
                float* p = feat + cx.srow + 32 * c + 4 * cx.g;
                st4g(p, acc[0]);
                st4g((p + 16), acc[1]);
            } else {
                sdf[cx.g == 0 ? cx.srow >> 8 : cx.pad] = acc[0][0];   // row 256 lives in the g = 0 lanes; the others hit the pad entry
            }
            return;
        }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = acc[e >> 2][e & 3] * K2;
            const float u = __builtin_amdgcn_exp2f(-__builtin_fabsf(t));
            x[e] = (__builtin_fmaxf(t, 0.0f) + __builtin_amdgcn_logf(1.0f + u)) * osc;
        }
        {
            float* p = xout + cx.srow + 32 * c + 4 * cx.g;
            st4g(p, (f32x4){x[0], x[1], x[2], x[3]});
            st4g((p + 16), (f32x4){x[4], x[5], x[6], x[7]});
        }
        if (c < 8) split8(x, Bn.h[c], Bn.l[c]);
    }
};

// gradient sweep: V_{l-1} = sigma'_{l-1} (.) T_l
struct EpiB {
    static constexpr bool NEXT_FROM_MEM = false;
    const float* xin;      // X_l rows: sigma'_{l-1} is derived from them
    float kx;
    float* uout;           // U_{l-1}
    float* vout;           // V_{l-1}
    bool final;            // the 39-row product with W_0^T: accumulators start from the Fourier rows of T_4, result -> G
    float* gout;           // [P][39]
    f32x4 px[2], nx[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        if (!final && c < 8) {
            nx[0] = ld4g(xin + cx.row + 32 * c + 4 * cx.g);
            nx[1] = ld4g(xin + cx.row + 32 * c + 16 + 4 * cx.g);
            return 2;
        }
        return 0;
    }
    __device__ __forceinline__ void rotate() { px[0] = nx[0]; px[1] = nx[1]; }
    __device__ __forceinline__ void touch() { touch4(px[0]); touch4(px[1]); }
    __device__ __forceinline__ void init(const Ctx& cx, int c, f32x4 (&acc)[2]) {
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (final && c < 2) {
            acc[0] = *(const f32x4*)(cx.binf + (c * 2 + 0) * TILE_B + cx.lane * 16);
            acc[1] = *(const f32x4*)(cx.binf + (c * 2 + 1) * TILE_B + cx.lane * 16);
        }
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        if (final) {
            if (cx.valid && c < 2) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 32 * c + 16 * mb + 4 * cx.g + i;
                        if (r < E_PE) gout[cx.prow * E_PE + r] = acc[mb][i];
                    }
            }
            return;
        }
        if (c >= 8) {   // layer 4's Fourier rows: parked in the wave's LDS block until the final product
            *(f32x4*)(cx.binf + ((c - 8) * 2 + 0) * TILE_B + cx.lane * 16) = acc[0];
            *(f32x4*)(cx.binf + ((c - 8) * 2 + 1) * TILE_B + cx.lane * 16) = acc[1];
            return;
        }
        float u[8], v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            u[e] = acc[e >> 2][e & 3];
            v[e] = (1.0f - one_minus_sig(px[e >> 2][e & 3], kx)) * u[e];
        }
        {
            float* pu = uout + cx.srow + 32 * c + 4 * cx.g;
            float* pv = vout + cx.srow + 32 * c + 4 * cx.g;
            st4g(pu, (f32x4){u[0], u[1], u[2], u[3]});
            st4g((pu + 16), (f32x4){u[4], u[5], u[6], u[7]});
            st4g(pv, (f32x4){v[0], v[1], v[2], v[3]});
            st4g((pv + 16), (f32x4){v[4], v[5], v[6], v[7]});
        }
        split8(v, Bn.h[c], Bn.l[c]);
    }
};

__global__ __launch_bounds__(TF_THREADS) void k_tf_sdf_fwd(TfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx;
    ctx_setup(cx, smem, a.wpack, a.P, FWD_CHUNKS, FWD_CHUNKS, 0, a.arena + 46 * (size_t)(a.P + 1) * HID + 117 * (size_t)a.P);
    const size_t R1 = (size_t)(a.P + 1) * HID;
    build_bin(cx, a.arena + off_IN(R1));
    tf_prologue(cx);
    BReg Bcur, Bnext;
#pragma unroll
    for (int k = 0; k < 8; ++k) { Bcur.h[k] = Bcur.l[k] = Bnext.h[k] = Bnext.l[k] = zero_frag(); }
    // ---- value sweep
    for (int l = 0; l <= 8; ++l) {
        EpiA ep;
        ep.bias = a.bias + l * BIAS_LD;
        ep.xout = a.arena + off_X(R1, l < 8 ? l + 1 : 8);
        ep.osc = (l == 3 ? R2 : 1.0f) / K2;
        ep.linear = l == 8;
        ep.feat = a.feat;
        ep.sdf = a.sdf;
        tf_layer<EpiA, 9, true>(cx, ep, l == 8 ? 9 : 8, l > 0, l == 0 || l == 4, Bcur, Bnext);
    }
    // ---- top of the gradient sweep: V_7 = sigma'_7 (.) w8, from X_8 (still the operand registers of layer 8)
    {
        float* vout = a.arena + off_V(R1, 7);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 w0 = ld4g(a.w8 + 32 * ks + 4 * cx.g), w1 = ld4g(a.w8 + 32 * ks + 16 + 4 * cx.g);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (float)Bcur.h[ks][e] + (float)Bcur.l[ks][e];
                v[e] = (1.0f - one_minus_sig(x, K2)) * (e < 4 ? w0[e & 3] : w1[e & 3]);
            }
            {
                float* pv = vout + cx.srow + 32 * ks + 4 * cx.g;
                st4g(pv, (f32x4){v[0], v[1], v[2], v[3]});
                st4g((pv + 16), (f32x4){v[4], v[5], v[6], v[7]});
            }
            split8(v, Bcur.h[ks], Bcur.l[ks]);
        }
    }
    // ---- gradient sweep
    for (int l = 7; l >= 0; --l) {
        EpiB ep;
        ep.final = l == 0;
        ep.xin = a.arena + off_X(R1, l > 0 ? l : 1);
        ep.kx = l == 4 ? K2 / R2 : K2;
        ep.uout = a.arena + off_U(R1, l > 0 ? l - 1 : 0);
        ep.vout = a.arena + off_V(R1, l > 0 ? l - 1 : 0);
        ep.gout = a.arena + off_G(R1, a.P);
        tf_layer<EpiB, 10, false>(cx, ep, l == 4 ? 10 : (l == 0 ? 2 : 8), true, false, Bcur, Bnext);
    }
}

// ================================================================================================================ value only
// The SAMPLER's network queries at the training path's arithmetic (round 5): the value sweep alone, sdf row only, nothing
// stashed -- split-bfloat16 products with fp32 accumulation and fp32 softplus, where the inference kernel k_mlp_sdf (csrc/mlp.hip)
// rounds inputs, weights and activations to half precision (each of the three contributes ~1e-4 to the sdf, measured).  The
// sampler's depths are an inverse CDF of these values and a shifted sample can cross the reference's `dist > 0.1 => sdf = 4`
// discontinuity (multiply.py:142-143): with these queries the depths agree with the fp32 oracle to 1e-3 instead of 2e-2 and the
// opacity of the worst grazing ray to 9e-4 instead of 1.1e-1 (profiles/r05_sampler_precision.txt).  Same chunk stream as the
// forward kernel's value sweep (chunks 0..63, then chunk 72 = the sdf row), same worklist interface as mp_mlp_sdf.
struct TfValArgs {
    const char* wpack;
    const float* bias;      // [9][288], layer 0 with the call's conditioning hoisted in (FusedSDFState.refresh)
    const float* xc;        // [*][3] canonical points
    const int* worklist;    // point ids, or NULL = identity
    const int* count_p;     // device-side number of work items (NULL: max_count)
    float* sdf_out;         // [*], written at the point ids
    int max_count;
};
struct EpiV {
    static constexpr bool NEXT_FROM_MEM = false;
    const float* bias;     // this layer's biases in pack-row order (the last layer: its single chunk's)
    float osc;
    bool linear;
    float* sdf_out;
    int id;                // this lane's point id (lanes of one column share it), -1 = none
    f32x4 pb[2], nb[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        nb[0] = ld4g(bias + 32 * c + 4 * cx.g);
        nb[1] = ld4g(bias + 32 * c + 16 + 4 * cx.g);
        return 2;
    }
    __device__ __forceinline__ void rotate() { pb[0] = nb[0]; pb[1] = nb[1]; }
    __device__ __forceinline__ void touch() { touch4(pb[0]); touch4(pb[1]); }
    __device__ __forceinline__ void init(const Ctx&, int, f32x4 (&acc)[2]) {
        acc[0] = pb[0];
        acc[1] = pb[1];
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        if (linear) {
            if (cx.g == 0 && id >= 0) sdf_out[id] = acc[0][0];     // row 256 of the last layer = the sdf row (chunk 72, row 0)
            return;
        }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = acc[e >> 2][e & 3] * K2;
            const float u = __builtin_amdgcn_exp2f(-__builtin_fabsf(t));
            x[e] = (__builtin_fmaxf(t, 0.0f) + __builtin_amdgcn_logf(1.0f + u)) * osc;
        }
        if (c < 8) split8(x, Bn.h[c], Bn.l[c]);
    }
};

__global__ __launch_bounds__(TF_THREADS) void k_tf_sdf_val(TfValArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int count = a.count_p ? min(*a.count_p, a.max_count) : a.max_count;
    for (int tile = blockIdx.x; tile * TF_PTS < count; tile += gridDim.x) {
        Ctx cx;
        ctx_setup(cx, smem, a.wpack, count, 65, 64, 8, nullptr);     // stream position 64 = chunk 72
        const int w = tile * TF_PTS + cx.wave * 16 + cx.j;
        const int id = w < count ? (a.worklist ? a.worklist[w] : w) : -1;
        {   // the wave's input-fed B fragments: Fourier features of its 16 points, natural slot order (39 of 64 used), as
            // k_pe_fwd computes them (sinf / cosf of x 2^k: the training forward's encoding)
            float px = 0.f, py = 0.f, pz = 0.f;
            if (id >= 0) { px = a.xc[3 * (size_t)id]; py = a.xc[3 * (size_t)id + 1]; pz = a.xc[3 * (size_t)id + 2]; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int f = 32 * ks + 8 * cx.g + e;
                    float val = 0.0f;
                    if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
                    else if (f < E_PE) {
                        const int k = (f - 3) / 6, r = (f - 3) % 6, ax = r % 3;
                        const float arg = (ax == 0 ? px : (ax == 1 ? py : pz)) * (float)(1 << k);
                        val = r < 3 ? sinf(arg) : cosf(arg);
                    }
                    v[e] = val;
                }
                bf16x8 hi, lo;
                split8(v, hi, lo);
                *(bf16x8*)(cx.binf + (ks * 2 + 0) * TILE_B + cx.lane * 16) = hi;
                *(bf16x8*)(cx.binf + (ks * 2 + 1) * TILE_B + cx.lane * 16) = lo;
            }
        }
        tf_prologue(cx);
        BReg Bcur, Bnext;
#pragma unroll
        for (int k = 0; k < 8; ++k) { Bcur.h[k] = Bcur.l[k] = Bnext.h[k] = Bnext.l[k] = zero_frag(); }
        for (int l = 0; l <= 8; ++l) {
            EpiV ep;
            ep.bias = a.bias + l * BIAS_LD + (l == 8 ? 256 : 0);
            ep.osc = (l == 3 ? R2 : 1.0f) / K2;
            ep.linear = l == 8;
            ep.sdf_out = a.sdf_out;
            ep.id = id;
#if defined(MP_DMA_SPLIT)
            tf_layer<EpiV, 8, true, 2>(cx, ep, l == 8 ? 1 : 8, l > 0, l == 0 || l == 4, Bcur, Bnext);
#elif !defined(MP_DMA_LATE)
            tf_layer<EpiV, 8, true, 1>(cx, ep, l == 8 ? 1 : 8, l > 0, l == 0 || l == 4, Bcur, Bnext);
#else
            tf_layer<EpiV, 8, true>(cx, ep, l == 8 ? 1 : 8, l > 0, l == 0 || l == 4, Bcur, Bnext);
#endif
        }
        __syncthreads();      // the ring and the input blocks are rebuilt by the next tile
    }
}

// ================================================================================================================ backward
// sum over the 16 points of a wave (lanes j = 0..15 of each group g) of 8 per-lane values -> LDS column sums
__device__ __forceinline__ void col_reduce(const Ctx& cx, int c, float (&r)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = r[e];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        r[e] = v;
    }
    if (cx.j == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(cx.redf + 32 * c + (e < 4 ? 4 * cx.g + e : 16 + 4 * cx.g + e - 4), r[e]);
    }
}

// ascending sweep: dV_l arrives in the accumulators;  dU = s (.) dV,  dS = s'' (.) U (.) dV,  dT_{l+1} = scale * dU
struct EpiC {
    // The next layer's operand (dT_{l+1}) is NOT kept in registers while this layer runs: it is re-read from the rows this wave
    // has just stored for the weight-gradient contraction (its own stores: in order, L2-resident).  This sweep prefetches four
    // stash vectors per chunk, double-buffered; with the 64 operand registers on top the kernel spilled, and a scratch reload in
    // the chunk loop drains the memory pipeline (vmcnt counts scratch traffic) -- exactly what the prefetch order is there to avoid.
    static constexpr bool NEXT_FROM_MEM = true;
    __device__ __forceinline__ void load_next(const Ctx& cx, BReg& B) {
        if (top) return;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 v0 = ld4g(dtout + cx.row + 32 * ks + 4 * cx.g), v1 = ld4g(dtout + cx.row + 32 * ks + 16 + 4 * cx.g);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            split8(v, B.h[ks], B.l[ks]);
        }
    }
    const float* xin;      // X_{l+1} rows
    float kx;
    const float* uin;      // U_l rows; top layer (7): the row vector w8
    float* dtout;          // dT_{l+1} rows (null at the top)
    float* dsout;          // dS_l rows
    float osc;
    bool top;
    f32x4 px[2], pu[2], nx[2], nu[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        const int col = 32 * c + 4 * cx.g;
        nx[0] = ld4g(xin + cx.row + col);
        nx[1] = ld4g(xin + cx.row + col + 16);
        if (top) {
            nu[0] = ld4g(uin + col);
            nu[1] = ld4g(uin + col + 16);
        } else {
            nu[0] = ld4g(uin + cx.row + col);
            nu[1] = ld4g(uin + cx.row + col + 16);
        }
        return 4;
    }
    __device__ __forceinline__ void rotate() { px[0] = nx[0]; px[1] = nx[1]; pu[0] = nu[0]; pu[1] = nu[1]; }
    __device__ __forceinline__ void touch() { touch4(px[0]); touch4(px[1]); touch4(pu[0]); touch4(pu[1]); }
    __device__ __forceinline__ void init(const Ctx&, int, f32x4 (&acc)[2]) {
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        float du[8], ds[8], dt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dv = acc[e >> 2][e & 3];
            const float q = one_minus_sig(px[e >> 2][e & 3], kx), s1 = 1.0f - q;
            du[e] = s1 * dv;
            ds[e] = 100.0f * s1 * q * pu[e >> 2][e & 3] * dv;
            dt[e] = du[e] * osc;
        }
        {
            float* ps = dsout + cx.srow + 32 * c + 4 * cx.g;
            st4g(ps, (f32x4){ds[0], ds[1], ds[2], ds[3]});
            st4g((ps + 16), (f32x4){ds[4], ds[5], ds[6], ds[7]});
        }
        if (top) {
            if (!cx.valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) du[e] = 0.0f;
            }
            col_reduce(cx, c, du);
            return;
        }
        {
            float* pt = dtout + cx.srow + 32 * c + 4 * cx.g;
            st4g(pt, (f32x4){dt[0], dt[1], dt[2], dt[3]});
            st4g((pt + 16), (f32x4){dt[4], dt[5], dt[6], dt[7]});
        }
    }
};

// descending sweep: dX_l arrives in the accumulators;  dZ_{l-1} = s_{l-1} (.) dX_l + dS_{l-1}
struct EpiD {
    static constexpr bool NEXT_FROM_MEM = true;      // see EpiC: the next operand dZ_{l-1} is re-read from the rows just stored
    __device__ __forceinline__ void load_next(const Ctx& cx, BReg& B) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 v0 = ld4g(dzout + cx.row + 32 * ks + 4 * cx.g), v1 = ld4g(dzout + cx.row + 32 * ks + 16 + 4 * cx.g);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            split8(v, B.h[ks], B.l[ks]);
        }
    }
    const float* xin;      // X_l rows
    float kx;
    const float* dsin;     // dS_{l-1} rows; NULL: a value-only network (no gradient sweep, hence no second-order term)
    float* dzout;          // dZ_{l-1} rows
    bool first;            // layer 8: accumulators start from w8 (x) d sdf (a rank-1 term); also sums d sdf . X_8
    const float* w8;
    float dsdf;
    f32x4 px[2], pd[2], nx[2], nd[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        const int col = 32 * c + 4 * cx.g;
        nx[0] = ld4g(xin + cx.row + col);
        nx[1] = ld4g(xin + cx.row + col + 16);
        if (dsin == nullptr) {
            nd[0] = nd[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            return 2;
        }
        nd[0] = ld4g(dsin + cx.row + col);
        nd[1] = ld4g(dsin + cx.row + col + 16);
        return 4;
    }
    __device__ __forceinline__ void rotate() { px[0] = nx[0]; px[1] = nx[1]; pd[0] = nd[0]; pd[1] = nd[1]; }
    __device__ __forceinline__ void touch() { touch4(px[0]); touch4(px[1]); touch4(pd[0]); touch4(pd[1]); }
    __device__ __forceinline__ void init(const Ctx& cx, int c, f32x4 (&acc)[2]) {
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (first) {
            acc[0] = ld4g(w8 + 32 * c + 4 * cx.g) * dsdf;
            acc[1] = ld4g(w8 + 32 * c + 16 + 4 * cx.g) * dsdf;
        }
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        float dz[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            dz[e] = (1.0f - one_minus_sig(px[e >> 2][e & 3], kx)) * acc[e >> 2][e & 3] + pd[e >> 2][e & 3];
        {
            float* pz = dzout + cx.srow + 32 * c + 4 * cx.g;
            st4g(pz, (f32x4){dz[0], dz[1], dz[2], dz[3]});
            st4g((pz + 16), (f32x4){dz[4], dz[5], dz[6], dz[7]});
        }
        if (first) {            // the value sweep's part of the sdf row's gradient: sum over points of d sdf . X_8
            float r[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = cx.valid ? dsdf * px[e >> 2][e & 3] : 0.0f;
            col_reduce(cx, c, r);
        }
    }
};

__global__ __launch_bounds__(TF_THREADS) void k_tf_sdf_bwd(TfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx;
    ctx_setup(cx, smem, a.wpack, a.P, BWD_CHUNKS, 64, BWD_SHIFT, a.arena + 46 * (size_t)(a.P + 1) * HID + 117 * (size_t)a.P);
    const size_t R1 = (size_t)(a.P + 1) * HID;
    build_bin(cx, a.arena + off_dG(R1, a.P));
    if (threadIdx.x < 257) cx.redf[threadIdx.x] = 0.0f;
    tf_prologue(cx);
    BReg Bcur, Bnext;
#pragma unroll
    for (int k = 0; k < 8; ++k) { Bcur.h[k] = Bcur.l[k] = Bnext.h[k] = Bnext.l[k] = zero_frag(); }
    // ---- ascending: the adjoint of the gradient sweep
    for (int l = 0; l <= 7; ++l) {
        EpiC ep;
        ep.xin = a.arena + off_X(R1, l + 1);
        ep.kx = l == 3 ? K2 / R2 : K2;
        ep.top = l == 7;
        ep.uin = l == 7 ? a.w8 : a.arena + off_U(R1, l);
        ep.dtout = a.arena + off_dT(R1, l < 7 ? l + 1 : 7);
        ep.dsout = a.arena + off_dS(R1, l);
        ep.osc = l == 3 ? R2 : 1.0f;
        tf_layer<EpiC, 8, true>(cx, ep, 8, l > 0, l == 0 || l == 4, Bcur, Bnext);
    }
    // ---- descending: the adjoint of the value sweep, from dZ_8 [P][257] = (d sdf | d features)
    const float dsdf = a.dsdf[cx.prow];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const f32x4 v0 = ld4g(a.dfeat + cx.row + 32 * ks + 4 * cx.g), v1 = ld4g(a.dfeat + cx.row + 32 * ks + 16 + 4 * cx.g);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(v, Bcur.h[ks], Bcur.l[ks]);
    }
    {   // the last layer's sdf bias gradient: sum over the workgroup's points of d sdf
        float t = (cx.valid && cx.g == 0) ? dsdf : 0.0f;
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        t += __shfl_xor(t, 8);
        if (cx.lane == 0) atomicAdd(cx.redf + 256, t);
    }
    for (int l = 8; l >= 1; --l) {
        EpiD ep;
        ep.xin = a.arena + off_X(R1, l);
        ep.kx = l == 4 ? K2 / R2 : K2;
        ep.dsin = a.arena + off_dS(R1, l - 1);
        ep.dzout = a.arena + off_dZ(R1, l - 1);
        ep.first = l == 8;
        ep.w8 = a.w8;
        ep.dsdf = dsdf;
        tf_layer<EpiD, 8, false>(cx, ep, 8, true, false, Bcur, Bnext);
    }
    __syncthreads();
    if (threadIdx.x < 256) atomicAdd(a.dw8 + threadIdx.x, cx.redf[threadIdx.x]);
    if (threadIdx.x == 256) atomicAdd(a.db8, cx.redf[256]);
}

// ================================================================================================================ background net
// The NeRF++ background ImplicitNet (networks.py:126-208 with confs/model: d_in 4, multires 10 -> 84 Fourier features, frame code
// (32) hoisted into layer 0's bias, 8 x 256 softplus, skip connection at layer 4 = [172 | 84] / sqrt 2, 257 outputs, no weight
// norm; multiply.py:514-541) on the same skeleton, VALUE ONLY: the training forward reads its density (|sdf|) and features, no
// spatial gradient, so the backward is the plain descending sweep.  Round 4 ran it layer by layer (9 GEMMs + 8 softplus passes
// forward, 3 GEMMs + a pass per layer backward: ~45 launches, ~0.9 ms per iteration).
// The 84 Fourier features do not fit the 64 input-fed K slots of a chunk; they ride in the REGISTER operand instead: layer 0's
// operand is the feature row itself (K slots 0..83), and layer 4's operand is layer 3's 172 outputs followed by the 84 features
// (times 1/sqrt 2) in slots 172..255 -- exactly 256.  No chunk of this network has an input-fed part.
// chunk stream (40 KiB chunks as above): 0..72 value orientation W_l, l = 0..8 (layer 3: rows >= 172 zero; layer 8: 256 feature
// rows, then the sdf row); 73..136 for the backward: W_8[1:]^T, then W_l^T for l = 7..1 (l = 4: only the 172 rows of X_4's
// network part, times 1/sqrt 2).
// stash (floats), P points, R1 = (P + 1) 256:  dZ(l) l = 0..7 at l R1 | X(l) l = 1..8 at (7 + l) R1 | IN [P][84] at 16 R1
constexpr int BG_E = 84, BG_OUT3 = 172, BG_IN0 = 116, BG_FWD = 73, BG_TOTAL = 137;
__host__ __device__ inline size_t bgoff_dZ(size_t R1, int l) { return (size_t)l * R1; }
__host__ __device__ inline size_t bgoff_X(size_t R1, int l) { return (size_t)(7 + l) * R1; }
__host__ __device__ inline size_t bgoff_IN(size_t R1) { return 16 * R1; }

__device__ float pack_value_bg(const float* const* __restrict__ W, int chunk, int r, int f, int s) {
    if (f < 0) return 0.f;                                // no input-fed part
    if (chunk < BG_FWD) {                                 // value orientation
        const int l = chunk == 72 ? 8 : chunk >> 3, R = chunk == 72 ? 256 + r : 32 * (chunk & 7) + r;
        int o;
        if (l < 8) { o = R; if (R >= (l == 3 ? BG_OUT3 : HID)) return 0.f; }
        else { if (R > 256) return 0.f; o = R < 256 ? R + 1 : 0; }
        if (l == 0) return f < BG_E ? W[0][(size_t)o * BG_IN0 + f] : 0.f;
        return W[l][(size_t)o * HID + f];                 // (layer 4: columns 172.. multiply the re-injected features)
    }
    const int k = chunk - BG_FWD, l = k < 8 ? 8 : 7 - (k - 8) / 8, R = 32 * (k % 8) + r;
    if (l == 8) return W[8][(size_t)(f + 1) * HID + R];   // W_8[1:]^T
    if (f >= (l == 3 ? BG_OUT3 : HID)) return 0.f;        // K = the layer's outputs
    if (l == 4) return R < BG_OUT3 ? W[4][(size_t)f * HID + R] * R2 : 0.f;
    return W[l][(size_t)f * HID + R];
}
__global__ __launch_bounds__(512) void k_tf_pack_bg(const float* const* __restrict__ W, const float* const* __restrict__ B,
                                                    __bf16* __restrict__ wpack, float* __restrict__ bias_all) {
    pack_chunk(W, wpack, pack_value_bg);
    const int l = blockIdx.x;
    if (l < 9)
        for (int r = threadIdx.x; r < BIAS_LD; r += 512) {
            float b = 0.f;
            if (l < 8) { if (r < (l == 3 ? BG_OUT3 : HID)) b = B[l][r]; }
            else if (r <= 256) b = B[8][r < 256 ? r + 1 : 0];
            bias_all[l * BIAS_LD + r] = b;
        }
}

// this lane's eight K slots of K step ks from a row of 84 features (slot order = feature order through slot_feature), times sc;
// slots at or beyond `from` take the feature f - from, the others keep `keep`
__device__ __forceinline__ void bg_feature_slots(const Ctx& cx, const float* __restrict__ in_row, int ks, int from, float sc,
                                                 BReg& B) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int f = slot_feature(ks, cx.g, e);
        const float keep = (float)B.h[ks][e] + (float)B.l[ks][e];
        v[e] = f >= from ? (f - from < BG_E ? in_row[f - from] * sc : 0.0f) : keep;
    }
    split8(v, B.h[ks], B.l[ks]);
}
__device__ __forceinline__ void bg_ctx(Ctx& cx) { cx.noreg_hi = 0; cx.in0_lo = cx.in0_hi = cx.in1_lo = cx.in1_hi = 0; }

__global__ __launch_bounds__(TF_THREADS) void k_tf_bg_fwd(TfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx;
    const size_t R1 = (size_t)(a.P + 1) * HID;
    ctx_setup(cx, smem, a.wpack, a.P, BG_FWD, BG_FWD, 0, nullptr);
    bg_ctx(cx);
    tf_prologue(cx);
    BReg Bcur, Bnext;
#pragma unroll
    for (int k = 0; k < 8; ++k) { Bcur.h[k] = Bcur.l[k] = Bnext.h[k] = Bnext.l[k] = zero_frag(); }
    const float* in_row = a.arena + bgoff_IN(R1) + cx.prow * BG_E;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) bg_feature_slots(cx, in_row, ks, 0, 1.0f, Bcur);        // layer 0's operand: the features
    for (int l = 0; l <= 8; ++l) {
        EpiA ep;
        ep.bias = a.bias + l * BIAS_LD;
        ep.xout = a.arena + bgoff_X(R1, l < 8 ? l + 1 : 8);
        ep.osc = (l == 3 ? R2 : 1.0f) / K2;
        ep.linear = l == 8;
        ep.feat = a.feat;
        ep.sdf = a.sdf;
        tf_layer<EpiA, 9, false>(cx, ep, l == 8 ? 9 : 8, true, false, Bcur, Bnext);
        if (l == 3) {                                      // the skip connection: slots 172..255 of layer 4's operand
#pragma unroll
            for (int ks = 5; ks < 8; ++ks) bg_feature_slots(cx, in_row, ks, BG_OUT3, R2, Bcur);
        }
    }
}

__global__ __launch_bounds__(TF_THREADS) void k_tf_bg_bwd(TfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx;
    const size_t R1 = (size_t)(a.P + 1) * HID;
    ctx_setup(cx, smem, a.wpack, a.P, 64, 0, BG_FWD, nullptr);       // stream position k = chunk 73 + k
    bg_ctx(cx);
    if (threadIdx.x < 257) cx.redf[threadIdx.x] = 0.0f;
    tf_prologue(cx);
    BReg Bcur, Bnext;
#pragma unroll
    for (int k = 0; k < 8; ++k) { Bnext.h[k] = Bnext.l[k] = zero_frag(); }
    const float dsdf = a.dsdf[cx.prow];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const f32x4 v0 = ld4g(a.dfeat + cx.row + 32 * ks + 4 * cx.g), v1 = ld4g(a.dfeat + cx.row + 32 * ks + 16 + 4 * cx.g);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(v, Bcur.h[ks], Bcur.l[ks]);
    }
    {   // the last layer's sdf bias gradient: sum over the workgroup's points of d sdf
        float t = (cx.valid && cx.g == 0) ? dsdf : 0.0f;
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        t += __shfl_xor(t, 8);
        if (cx.lane == 0) atomicAdd(cx.redf + 256, t);
    }
    for (int l = 8; l >= 1; --l) {
        EpiD ep;
        ep.xin = a.arena + bgoff_X(R1, l);
        ep.kx = l == 4 ? K2 / R2 : K2;
        ep.dsin = nullptr;
        ep.dzout = a.arena + bgoff_dZ(R1, l - 1);
        ep.first = l == 8;
        ep.w8 = a.w8;
        ep.dsdf = dsdf;
        tf_layer<EpiD, 8, false>(cx, ep, 8, true, false, Bcur, Bnext);
    }
    __syncthreads();
    if (threadIdx.x < 256) atomicAdd(a.dw8 + threadIdx.x, cx.redf[threadIdx.x]);
    if (threadIdx.x == 256) atomicAdd(a.db8, cx.redf[256]);
}

// ================================================================================================================ colour net
// RenderingNet, mode 'pose_no_view' (networks.py:263-312): [x_c, n (6: input-fed) | pose embedding (8: hoisted into the bias) |
// features (256: register-fed, read from the SDF kernel's feat rows)] -> 4 x (256, ReLU) -> 3 -> sigmoid.  Same skeleton, ReLU.
// chunk stream: 0..7 layer 0; 8..31 layers 1..3; 32 layer 4 (3 rows);  33..56 W_l^T for l = 3, 2, 1; 57..64 W_0[:, 14:270]^T
// (rows = the 256 feature columns); 65 W_0[:, 0:6]^T (6 rows).  Forward: 0..32; backward: 33..65.
constexpr int COL_FWD = 33, COL_TOTAL = 66, COL_IN = 6, COL_FEAT0 = 14, COL_K0 = 270;
__device__ float pack_value_col(const float* const* __restrict__ W, int chunk, int r, int f, int s) {
    const bool reg = f >= 0;
    if (chunk < COL_FWD) {
        const int l = chunk < 32 ? chunk >> 3 : 4, R = chunk < 32 ? 32 * (chunk & 7) + r : r;
        if (R >= (l < 4 ? HID : 3)) return 0.f;
        if (reg) return l == 0 ? W[0][(size_t)R * COL_K0 + COL_FEAT0 + f] : W[l][(size_t)R * HID + f];
        return (l == 0 && s < COL_IN) ? W[0][(size_t)R * COL_K0 + s] : 0.f;
    }
    if (!reg) return 0.f;
    if (chunk < 57) { const int l = 3 - (chunk - 33) / 8, R = 32 * ((chunk - 33) % 8) + r; return W[l][(size_t)f * HID + R]; }
    if (chunk < 65) { const int R = 32 * (chunk - 57) + r; return W[0][(size_t)f * COL_K0 + COL_FEAT0 + R]; }
    return r < COL_IN ? W[0][(size_t)f * COL_K0 + r] : 0.f;
}
__global__ __launch_bounds__(512) void k_tf_pack_col(const float* const* __restrict__ W, const float* const* __restrict__ B,
                                                     __bf16* __restrict__ wpack, float* __restrict__ bias_all) {
    pack_chunk(W, wpack, pack_value_col);
    const int l = blockIdx.x;
    if (l < 5)
        for (int r = threadIdx.x; r < BIAS_LD; r += 512) bias_all[l * BIAS_LD + r] = r < (l < 4 ? HID : 3) ? B[l][r] : 0.f;
}

// stash (floats), n points, N1 = (n + 1) * 256 (pad rows as above): H(l) l = 0..3 at l N1 (layer l's ReLU output); dZ(l) at (4 + l) N1
struct ColArgs {
    const char* wpack;
    const float* bias;     // [5][288]
    float* stash;
    const float* feat;     // [n][256]
    const float* xa;       // [n][6]
    float* rgb;            // fwd out / bwd in: [n][3]
    const float* drgb;     // bwd: [n][3]
    const float* w4;       // bwd: W_4 [3][256]
    float* dfeat;          // bwd: [n][256]
    float* dxa;            // bwd: [n][6]
    float* dz4;            // bwd: [n][3]  (adjoint of the last layer's pre-activations, for its weight gradient)
    int n;
};

struct EpiR {
    static constexpr bool NEXT_FROM_MEM = false;
    const float* bias;
    float* hout;
    bool last;
    float* rgb;
    f32x4 pb[2], nb[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        nb[0] = ld4g(bias + 32 * c + 4 * cx.g);
        nb[1] = ld4g(bias + 32 * c + 16 + 4 * cx.g);
        return 2;
    }
    __device__ __forceinline__ void rotate() { pb[0] = nb[0]; pb[1] = nb[1]; }
    __device__ __forceinline__ void touch() { touch4(pb[0]); touch4(pb[1]); }
    __device__ __forceinline__ void init(const Ctx&, int, f32x4 (&acc)[2]) {
        acc[0] = pb[0];
        acc[1] = pb[1];
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        if (last) {
            if (cx.valid && c == 0 && cx.g == 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) rgb[cx.prow * 3 + a] = 1.0f / (1.0f + expf(-acc[0][a]));
            }
            return;
        }
        float h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = __builtin_fmaxf(acc[e >> 2][e & 3], 0.0f);
        {
            float* p = hout + cx.srow + 32 * c + 4 * cx.g;
            st4g(p, (f32x4){h[0], h[1], h[2], h[3]});
            st4g((p + 16), (f32x4){h[4], h[5], h[6], h[7]});
        }
        if (c < 8) split8(h, Bn.h[c], Bn.l[c]);
    }
};

__global__ __launch_bounds__(TF_THREADS) void k_tf_col_fwd(ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx;
    ctx_setup(cx, smem, a.wpack, a.n, COL_FWD, COL_FWD, 0, a.stash + (size_t)(a.n + 1) * 8 * HID);
    cx.noreg_hi = 0; cx.in0_lo = 0; cx.in0_hi = 8; cx.in1_lo = cx.in1_hi = 0;
    const size_t NL = (size_t)(a.n + 1) * HID;
    build_bin(cx, a.xa, COL_IN);
    tf_prologue(cx);
    BReg Bcur, Bnext;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const f32x4 v0 = ld4g(a.feat + cx.row + 32 * ks + 4 * cx.g), v1 = ld4g(a.feat + cx.row + 32 * ks + 16 + 4 * cx.g);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(v, Bcur.h[ks], Bcur.l[ks]);
        Bnext.h[ks] = Bnext.l[ks] = zero_frag();
    }
    for (int l = 0; l <= 4; ++l) {
        EpiR ep;
        ep.bias = a.bias + l * BIAS_LD;
        ep.hout = a.stash + (size_t)(l < 4 ? l : 3) * NL;
        ep.last = l == 4;
        ep.rgb = a.rgb;
        tf_layer<EpiR, 8, true>(cx, ep, l == 4 ? 1 : 8, true, l == 0, Bcur, Bnext);
    }
}

// backward: dH_{l-1} arrives in the accumulators; dZ_{l-1} = [H_{l-1} > 0] dH_{l-1};  last product: d feat (256 rows) and d XA (6)
struct EpiRb {
    static constexpr bool NEXT_FROM_MEM = false;
    const float* hin;
    float* dzout;
    bool final;
    float* dfeat;
    float* dxa;
    f32x4 ph[2], nh[2];
    __device__ __forceinline__ int prefetch(const Ctx& cx, int c) {
        if (!final) {
            nh[0] = ld4g(hin + cx.row + 32 * c + 4 * cx.g);
            nh[1] = ld4g(hin + cx.row + 32 * c + 16 + 4 * cx.g);
            return 2;
        }
        return 0;
    }
    __device__ __forceinline__ void rotate() { ph[0] = nh[0]; ph[1] = nh[1]; }
    __device__ __forceinline__ void touch() { touch4(ph[0]); touch4(ph[1]); }
    __device__ __forceinline__ void init(const Ctx&, int, f32x4 (&acc)[2]) {
        acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void run(const Ctx& cx, int c, const f32x4 (&acc)[2], BReg& Bn) {
        if (final) {
            if (c < 8) {
                float* p = (cx.valid ? dfeat + cx.row : cx.trash) + 32 * c + 4 * cx.g;   // caller-owned rows: no pad row
                st4g(p, acc[0]);
                st4g((p + 16), acc[1]);
            } else if (cx.valid) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * cx.g + i < COL_IN) dxa[cx.prow * COL_IN + 4 * cx.g + i] = acc[0][i];
            }
            return;
        }
        float dz[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) dz[e] = ph[e >> 2][e & 3] > 0.0f ? acc[e >> 2][e & 3] : 0.0f;
        {
            float* p = dzout + cx.srow + 32 * c + 4 * cx.g;
            st4g(p, (f32x4){dz[0], dz[1], dz[2], dz[3]});
            st4g((p + 16), (f32x4){dz[4], dz[5], dz[6], dz[7]});
        }
        if (c < 8) split8(dz, Bn.h[c], Bn.l[c]);
    }
};

__global__ __launch_bounds__(TF_THREADS) void k_tf_col_bwd(ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Ctx cx;
    ctx_setup(cx, smem, a.wpack, a.n, COL_TOTAL - COL_FWD, 0, COL_FWD, a.stash + (size_t)(a.n + 1) * 8 * HID);
    cx.noreg_hi = 0; cx.in0_lo = cx.in0_hi = cx.in1_lo = cx.in1_hi = 0;
    const size_t NL = (size_t)(a.n + 1) * HID;
    tf_prologue(cx);
    BReg Bcur, Bnext;
    // the sigmoid and the 3-row last layer by hand: dz4 = d rgb . rgb (1 - rgb);  dH_3 = W_4^T dz4;  dZ_3 = [H_3 > 0] dH_3
    float dz4[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float y = a.rgb[cx.prow * 3 + k];
        dz4[k] = a.drgb[cx.prow * 3 + k] * y * (1.0f - y);
    }
    if (cx.valid && cx.g == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) a.dz4[cx.prow * 3 + k] = dz4[k];
    }
    {
        const float* h3 = a.stash + 3 * NL + cx.row;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int col = 32 * ks + 4 * cx.g;
            const f32x4 h0 = ld4g(h3 + col), h1 = ld4g(h3 + col + 16);
            f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d0 += ld4g(a.w4 + k * HID + col) * dz4[k];
                d1 += ld4g(a.w4 + k * HID + col + 16) * dz4[k];
            }
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = h0[e] > 0.0f ? d0[e] : 0.0f; v[4 + e] = h1[e] > 0.0f ? d1[e] : 0.0f; }
            {
                float* pd = a.stash + 7 * NL + cx.srow + col;
                st4g(pd, (f32x4){v[0], v[1], v[2], v[3]});
                st4g((pd + 16), (f32x4){v[4], v[5], v[6], v[7]});
            }
            split8(v, Bcur.h[ks], Bcur.l[ks]);
            Bnext.h[ks] = Bnext.l[ks] = zero_frag();
        }
    }
    for (int l = 3; l >= 0; --l) {
        EpiRb ep;
        ep.final = l == 0;
        ep.hin = a.stash + (size_t)(l > 0 ? l - 1 : 0) * NL;
        ep.dzout = a.stash + (size_t)(4 + (l > 0 ? l - 1 : 0)) * NL;
        ep.dfeat = a.dfeat;
        ep.dxa = a.dxa;
        tf_layer<EpiRb, 9, false>(cx, ep, l == 0 ? 9 : 8, true, false, Bcur, Bnext);
    }
}

}  // namespace

extern "C" int mp_tf_sdf_sizes(int P, long long* arena_floats, long long* pack_bytes) {
    if (arena_floats) *arena_floats = 46LL * (P + 1) * HID + 117LL * P + 256;
    if (pack_bytes) *pack_bytes = (long long)CH_TOTAL * CH_BYTES;
    return 0;
}

extern "C" int mp_tf_sdf_pack(const float* const* W, const float* const* B, void* wpack, float* bias_all, void* stream) {
    hipLaunchKernelGGL(k_tf_pack, dim3(CH_TOTAL), dim3(512), 0, (hipStream_t)stream, W, B, (__bf16*)wpack, bias_all);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_sdf_fwd(const void* wpack, const float* bias_all, const float* w8, float* arena, int P, float* feat,
                             float* sdf, void* stream) {
    if (P <= 0) return 0;
    MP_LDS_ATTR(k_tf_sdf_fwd, LDS_BYTES);
    TfArgs a{(const char*)wpack, bias_all, w8, arena, feat, sdf, nullptr, nullptr, nullptr, nullptr, P};
    hipLaunchKernelGGL(k_tf_sdf_fwd, dim3((P + TF_PTS - 1) / TF_PTS), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_sdf_val(const void* wpack, const float* bias_all, const float* xc, const int* worklist, const int* count,
                             int max_count, float* sdf_out, void* stream) {
    if (max_count <= 0) return 0;
    MP_LDS_ATTR(k_tf_sdf_val, LDS_BYTES);
    TfValArgs a{(const char*)wpack, bias_all, xc, worklist, count, sdf_out, max_count};
    const int tiles = (max_count + TF_PTS - 1) / TF_PTS;
    hipLaunchKernelGGL(k_tf_sdf_val, dim3(tiles < 256 ? tiles : 256), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_sdf_bwd(const void* wpack, const float* w8, float* arena, int P, const float* dfeat, const float* dsdf,
                             float* dw8, float* db8, void* stream) {
    if (P <= 0) return 0;
    MP_LDS_ATTR(k_tf_sdf_bwd, LDS_BYTES);
    TfArgs a{(const char*)wpack, nullptr, w8, arena, nullptr, nullptr, dfeat, dsdf, dw8, db8, P};
    hipLaunchKernelGGL(k_tf_sdf_bwd, dim3((P + TF_PTS - 1) / TF_PTS), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_bg_sizes(int P, long long* arena_floats, long long* pack_bytes) {
    if (arena_floats) *arena_floats = 16LL * (P + 1) * HID + (long long)BG_E * P + 256;
    if (pack_bytes) *pack_bytes = (long long)BG_TOTAL * CH_BYTES;
    return 0;
}

extern "C" int mp_tf_bg_pack(const float* const* W, const float* const* B, void* wpack, float* bias_all, void* stream) {
    hipLaunchKernelGGL(k_tf_pack_bg, dim3(BG_TOTAL), dim3(512), 0, (hipStream_t)stream, W, B, (__bf16*)wpack, bias_all);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_bg_fwd(const void* wpack, const float* bias_all, float* arena, int P, float* feat, float* sdf, void* stream) {
    if (P <= 0) return 0;
    MP_LDS_ATTR(k_tf_bg_fwd, LDS_BYTES);
    TfArgs a{(const char*)wpack, bias_all, nullptr, arena, feat, sdf, nullptr, nullptr, nullptr, nullptr, P};
    hipLaunchKernelGGL(k_tf_bg_fwd, dim3((P + TF_PTS - 1) / TF_PTS), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_bg_bwd(const void* wpack, const float* w8, float* arena, int P, const float* dfeat, const float* dsdf,
                            float* dw8, float* db8, void* stream) {
    if (P <= 0) return 0;
    MP_LDS_ATTR(k_tf_bg_bwd, LDS_BYTES);
    TfArgs a{(const char*)wpack, nullptr, w8, arena, nullptr, nullptr, dfeat, dsdf, dw8, db8, P};
    hipLaunchKernelGGL(k_tf_bg_bwd, dim3((P + TF_PTS - 1) / TF_PTS), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_col_sizes(int n, long long* stash_floats, long long* pack_bytes) {
    if (stash_floats) *stash_floats = 8LL * (n + 1) * HID + 256;
    if (pack_bytes) *pack_bytes = (long long)COL_TOTAL * CH_BYTES;
    return 0;
}

extern "C" int mp_tf_col_pack(const float* const* W, const float* const* B, void* wpack, float* bias_all, void* stream) {
    hipLaunchKernelGGL(k_tf_pack_col, dim3(COL_TOTAL), dim3(512), 0, (hipStream_t)stream, W, B, (__bf16*)wpack, bias_all);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_col_fwd(const void* wpack, const float* bias_all, float* stash, const float* feat, const float* xa, int n,
                             float* rgb, void* stream) {
    if (n <= 0) return 0;
    MP_LDS_ATTR(k_tf_col_fwd, LDS_BYTES);
    ColArgs a{(const char*)wpack, bias_all, stash, feat, xa, rgb, nullptr, nullptr, nullptr, nullptr, nullptr, n};
    hipLaunchKernelGGL(k_tf_col_fwd, dim3((n + TF_PTS - 1) / TF_PTS), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int mp_tf_col_bwd(const void* wpack, float* stash, const float* w4, const float* rgb, const float* drgb, int n,
                             float* dfeat, float* dxa, float* dz4, void* stream) {
    if (n <= 0) return 0;
    MP_LDS_ATTR(k_tf_col_bwd, LDS_BYTES);
    ColArgs a{(const char*)wpack, nullptr, stash, nullptr, nullptr, (float*)rgb, drgb, w4, dfeat, dxa, dz4, n};
    hipLaunchKernelGGL(k_tf_col_bwd, dim3((n + TF_PTS - 1) / TF_PTS), dim3(TF_THREADS), LDS_BYTES, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
