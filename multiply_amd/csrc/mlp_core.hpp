// Register-resident fused MLP core for gfx950 (CDNA4), f16 MFMA 16x16x32 (same rate as bf16, 10-bit mantissa), fp32
// accumulate.
//
// Replaces the per-layer torch.nn.Linear + Softplus/ReLU chain of the reference's
// ImplicitNet.forward (code/lib/model/networks.py:160-181) and RenderingNet.forward
// (networks.py:305-311): all layers of one network are evaluated for a tile of points
// without the activations ever leaving the register file.
//
// Orientation: D[out_feature][point] = W[out][k] * X^T[k][point].
//   A operand = 16x32 weight tile (rows = output features), streamed global -> LDS
//               (global_load_lds, 16 B / lane) in MFMA fragment order, shared by all
//               waves of the workgroup.
//   B operand = activations, 32 k-slots x 16 points per (ks, nb); lives in VGPRs.
//   D         = col = lane&15 (point), row = 4*(lane>>4)+reg (output feature).
// Because lane (j,g) receives output rows 4g..4g+3 of every 16-row block and needs
// k-slots 8g..8g+7 of every 32-slot K step of the next layer, the host packs the
// weights with the K permutation
//     slot(ks, g, e) <-> feature 32*ks + (e<4 ? 4g+e : 16+4g+(e-4))
// (hip.py reg_slot_feature / csrc/pack.hip mp_pack_layer), so that half(act(D)) of output blocks
// (2ks, 2ks+1) IS the B fragment of K step ks: no LDS round trip, no cross-lane moves.
//
// A wave owns NB column blocks of 16 columns.  Plain mode: 16*NB different points (NB = 2 with 8 waves per workgroup
// = 2 waves per SIMD, so that one wave's activation VALU work overlaps the other wave's MFMAs).
// Forward mode (FWD): value and d/dx, d/dy, d/dz tangent columns of the same points ride through the network together,
// t' = softplus'(z) * (W t), so sdf and its spatial gradient (the normals of multiply.py:620-661) come out of one pass.
// Layout: 8 points per wave in half blocks, block 0 = [values | d/dx], block 1 = [d/dy | d/dz].  (A 4-block layout, 16
// points per wave at one wave per SIMD, needs half the LDS reads and activation instructions per MFMA but was slower:
// 26.5 vs 24.0 ms per 4 M points -- a single wave cannot hide the latencies.)  The default shading path is reverse mode
// (mlp.hip: k_mlp_fwdsave + k_mlp_grad), which needs no tangent columns at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mp {

// MFMA operand element: IEEE half.  Same matrix-core rate as bf16 on gfx950, 3 more mantissa bits, and -- the reason it is
// used -- the activation code can run on PACKED pairs (v_pk_*_f16: two rows per instruction): this kernel family is bound
// by the number of VALU issue slots between MFMAs.  Range: hidden values carry the factor K = 100 log2(e) = 144 (scaled
// units, below), so |z| < 65504 / 144 = 454; tangent columns carry K * TANGENT_SCALE.
typedef _Float16 op_t;
typedef op_t opx8 __attribute__((ext_vector_type(8)));
typedef op_t h2 __attribute__((ext_vector_type(2)));
constexpr float TANGENT_SCALE = 0.0625f;   // forward-mode tangent columns are carried at 1/16 (undone on output)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KS_REG = 8;   // K steps (32 slots each) fed from registers = previous layer output (<=256 feats)
// KS_IN (template parameter, 2 or 3): K steps fed from the encoded network input (<=64 / 96 feats: PE, normals ...)
constexpr int TILE_BYTES = 1024;                  // one 16x32 half A tile, fragment order: [lane][8]
constexpr int CHUNK_MB = 2;                       // 32 output rows = one K step of the next layer
constexpr int MAX_CHUNKS = 9;                     // 8 chunks = 256 rows, +1 "extra output" chunk
constexpr int MAX_LAYERS = 10;
constexpr int BIAS_STRIDE = MAX_CHUNKS * 32;      // 288 floats per layer
__host__ __device__ constexpr int mb_bytes(int ks_in) { return (KS_REG + ks_in) * TILE_BYTES; }  // all K steps of 16 rows
__host__ __device__ constexpr int chunk_bytes(int ks_in) { return CHUNK_MB * mb_bytes(ks_in); }  // 20 / 22 KiB
constexpr int RING_SLOTS = 3;   // weight chunks are loaded two chunks ahead of their use
__host__ __device__ constexpr int in_stride(int ks_in) { return ks_in * 32 + 8; }  // halves per staging row (+pad)

enum Act : int { ACT_NONE = 0, ACT_SOFTPLUS = 1, ACT_RELU = 2, ACT_SIGMUL = 3 };

struct LayerDesc {
    int n_chunk;    // chunks of 32 output rows (1..9)
    int use_reg;    // consume the 8 register K steps
    int use_in;     // consume the 2 input K steps
    int act;        // Act
    int out_chunk;  // chunk whose first 16 rows are returned in fp32 `out` instead of feeding the next layer (-1: none)
    int aux;        // bits 0..7: 1 + index of the stored-sigmoid layer this layer's outputs are multiplied by (ACT_SIGMUL);
                    // bits 8..15: capture id (reverse sweep: which 48 output rows are the input-encoding gradient), 0 = none
};

struct NetDesc {
    int n_layers;
    int total_chunks;
    LayerDesc layer[MAX_LAYERS];
};

// One LDS-DMA piece: 64 lanes x 16 B global -> 1 KiB of LDS at `lds_base` (wave-uniform byte offset), asynchronous,
// counted in vmcnt.  Issued through inline asm ON PURPOSE: hipcc models the builtin (__builtin_amdgcn_global_load_lds)
// as a FLAT operation touching both memory and LDS, and while one is pending every wait for an LDS read degrades to a
// full drain (s_waitcnt lgkmcnt(0)) -- with weight chunks in flight all the time that is every wait of the K-step
// stream, and it defeats the A-tile prefetch queue.  The price: the compiler does not know these loads, so the code
// that consumes a chunk must order itself: dma_wait_all() before the barrier that publishes the chunk.
// Addressing: scalar 64-bit base (wave-uniform) + one 32-bit lane offset register shared by every piece.
__device__ __forceinline__ void lds_dma_16(const void* gsrc_uniform, unsigned lane_off, unsigned lds_base) {
    // (M0 is a reserved register for hipcc: it writes M0 itself right before every instruction of its own that reads it
    // and keeps nothing there across statements, so the asm may overwrite it without declaring a clobber)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base), "v"(lane_off), "s"(gsrc_uniform)
                 : "memory");
}
// s_waitcnt vmcnt(0) (gfx9 encoding: expcnt 7 and lgkmcnt 15 = "do not wait").  The builtin, not asm text: hipcc's wait
// insertion pass reads it and learns that ITS OWN global loads (inputs, stored sigmoids) are complete as well, so it
// does not re-wait for them (with vmcnt(0), i.e. also for the weight stream) inside the next chunk.
#ifdef MP_EXP_NOWAIT   // ablation (timing only, results are wrong): nobody waits for the weight DMA
__device__ __forceinline__ void dma_wait_all() {}
#else
__device__ __forceinline__ void dma_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }
#endif
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(size_t)p; }   // low half of the flat address
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const size_t v = (size_t)p;
    return (const char*)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

// Pieces [P0, P1) of this wave's share of weight chunk ci (a chunk = NP 1 KiB pieces dealt round-robin to the waves).
template <int KS_IN, int WAVES, int P0 = 0, int P1 = 99>
__device__ __forceinline__ void issue_chunk(const char* __restrict__ wpack, char* wring, int ci, int wave, int lane) {
    constexpr int CB = chunk_bytes(KS_IN);
    constexpr int NP = CB / TILE_BYTES;
    constexpr int NI = (NP + WAVES - 1) / WAVES;
    // everything the asm takes in scalar registers is made uniform explicitly ("s" does not enforce it)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave), ci_u = __builtin_amdgcn_readfirstlane(ci);
    const char* src = uniform_ptr(wpack) + (size_t)ci_u * CB + wave_u * TILE_BYTES;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_offset(wring)) + (ci_u % RING_SLOTS) * CB + wave_u * TILE_BYTES;
#pragma unroll
    for (int i = P0; i < (P1 < NI ? P1 : NI); ++i) {
        if (wave_u + WAVES * i < NP) lds_dma_16(src + WAVES * i * TILE_BYTES, lane * 16, dst + WAVES * i * TILE_BYTES);
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Weight DMA of the phase-separated stream: issued by the EARLY waves at the tail of their V phase (round 6, default) or -- -DMP_DMA_LATE,
// rounds 2-5 -- by the late waves at the head of theirs.  Same-box A/B (profiles/r06_early_dma_ab.txt): sampler SDF -2.6 %, forward
// sweep -2 %, colour -1 %, reverse sweep 0, mp_tf_sdf_val -3...5 %.
#if !defined(MP_DMA_LATE) && !defined(MP_DMA_EARLY)
#define MP_DMA_EARLY 1
#endif
#ifdef MP_EXP_NOMFMA   // ablation (timing only, results are wrong): no matrix instructions
#define MP_MFMA_F16(a, b, c, x, y, z) (c)
#else
#define MP_MFMA_F16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#endif

#ifdef MP_EXP_STAMP   // ablation tooling: shader-clock stamps of workgroup 0 at every chunk boundary (mp_debug_stamps)
__device__ unsigned long long mp_stamps[8 * 128 * 4];
#ifndef MP_STAMP_HID   // -DMP_STAMP_HID=<Hidden value>: only the kernels of that activation kind stamp (0 softplus, 1 relu,
#define MP_STAMP_HID -1   // 2 softplus + stored sigmoids, 3 reverse sweep); default: every kernel, the last one launched survives
#endif
#define MP_STAMP_OK(hid) (MP_STAMP_HID < 0 || (hid) == MP_STAMP_HID)
#define MP_STAMP(ev) do { if (MP_STAMP_OK(HID) && blockIdx.x == 0 && lane == 0 && ci < 120) mp_stamps[((wave) * 128 + ci) * 4 + (ev)] = __builtin_amdgcn_s_memtime(); } while (0)
// tile-level stamps in slots 120..127 of the same table (tools/tile_timeline.py)
#define MP_STAMP_AT(hid, slot, ev) do { if (MP_STAMP_OK(hid) && blockIdx.x == 0 && (threadIdx.x & 63) == 0) mp_stamps[((threadIdx.x >> 6) * 128 + (slot)) * 4 + (ev)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MP_STAMP(ev)
#define MP_STAMP_AT(hid, slot, ev)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 to_h2(float a, float b) {   // ONE v_cvt_pk_f16_f32 (RTNE): a vector conversion, so that it
    return __builtin_convertvector((f32x2){a, b}, h2);    // stays packed whatever consumes the result
}
__device__ __forceinline__ unsigned bits(h2 v) { return __builtin_bit_cast(unsigned, v); }

// Next-layer K operand under construction.  With NB = 4 the live state (Bcur 128 + Bnext 128 + input 32 + accumulators)
// exceeds the 256 architectural VGPRs; left to itself hipcc parks arbitrary pieces in AGPRs and pays a
// v_accvgpr_read for every MFMA operand.  Bnext is written once and read once per layer, so it is pinned in the
// accumulator file explicitly (one write and one read per register per layer) and everything hot stays in VGPRs.
template <int NB, bool IN_AGPR>
struct NextB {
    unsigned r[KS_REG][NB][4];
    // rows (2 j, 2 j + 1) of half-block `half` of K step c, already packed
    __device__ __forceinline__ void put(int c, int nb, int half, int j, h2 v) {
        const unsigned u = bits(v);
        if constexpr (IN_AGPR) {
            asm("v_accvgpr_write_b32 %0, %1" : "=a"(r[c][nb][2 * half + j]) : "v"(u));
        } else {
            r[c][nb][2 * half + j] = u;
        }
    }
    // same with a RUN-TIME K step (the layer's last block): a chain of selects, never a dynamically indexed array
    __device__ __forceinline__ void put_sel(int c_rt, int nb, int half, int j, h2 v) {
        static_assert(!IN_AGPR, "put_sel: VGPR-resident only");
        const unsigned u = bits(v);
#pragma unroll
        for (int c = 0; c < KS_REG; ++c) r[c][nb][2 * half + j] = c == c_rt ? u : r[c][nb][2 * half + j];
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int c = 0; c < KS_REG; ++c)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (IN_AGPR) asm("v_accvgpr_write_b32 %0, 0" : "=a"(r[c][nb][i]));
                    else r[c][nb][i] = 0u;
                }
    }
    __device__ __forceinline__ opx8 get(int c, int nb) const {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (IN_AGPR) {
                unsigned t;
                asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(r[c][nb][i]));
                v[i] = t;
            } else {
                v[i] = r[c][nb][i];
            }
        }
        return __builtin_bit_cast(opx8, v);
    }
};

// HID_SOFTPLUS_SAVE: softplus, and the sigmoid of every hidden unit is written out in the operand-fragment layout
//   [layer][K step][column block][lane][8 halves]  (one 16 B store per lane when a K step's operand is complete)
// HID_SIGMUL: the "activation" is a multiplication by such a stored sigmoid: the reverse sweep of reverse-mode
//   differentiation runs through the same core with the transposed weights.
// HID_SOFTPLUS_X2 (round 6): "split activations" -- the wave's two column blocks are the HIGH and the LOW half-precision part of
//   the SAME 16 points' activations (x = hi + lo, |lo| <= 2^-11 |x|: 22 mantissa bits), both multiplied by the same half-precision
//   weight tile: W_h x_h + W_h x_l, two MFMAs per product, accumulated in fp32; the softplus runs in fp32 on the sum of the two
//   accumulators and is split again.  What is left of the half-precision kernel's error is the rounding of the WEIGHTS, a tenth of
//   it (tools/sdf_split_study.py: the activation rounding is coherent from layer to layer, the weight rounding averages out).
enum Hidden : int { HID_SOFTPLUS = 0, HID_RELU = 1, HID_SOFTPLUS_SAVE = 2, HID_SIGMUL = 3, HID_SOFTPLUS_X2 = 4 };
struct SigIO {
    char* base;       // this wave's sigmoid block of the current tile
    int layer_bytes;  // bytes per layer in it (= 8 * NB * 1024)
};

// Softplus networks are evaluated in SCALED UNITS: every hidden pre-activation / activation carries the factor
// K = 100 log2(e) (the host scales biases and input-fed weights by K and the last, linear layer's weights by 1/K, see
// hip.py), because  K * softplus_100(z) = max(z',0) + log2(1 + 2^-|z'|)  with z' = K z:  base-2 softplus needs no
// multiplications around the two transcendentals, and d softplus/dz = sigmoid(100 z) = 2^(z' - h') is unit-free.
// (torch's threshold branch, 100 z > 20 -> z, is dropped: there the correction is below half an ulp of z.)
//
// The fp32 accumulators of two rows are rounded to a packed half pair FIRST (one v_cvt_pk_f16_f32) and the whole
// activation runs on the pair; its result is the next layer's operand register as it stands.
// v_exp_f16 / v_log_f16 have no packed form: low half, then high half written in place (SDWA, UNUSED_PRESERVE).  The
// s_nop 0 between them is the gfx940+ transcendental-result hazard: the second instruction READS the first one's result
// (to preserve the low half) and the assembler does not insert wait states inside an asm block.
__device__ __forceinline__ h2 exp2_h2(h2 x) {
    unsigned r;
    const unsigned xi = bits(x);
    asm("v_exp_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 0\n\t"
        "v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
        : "=&v"(r) : "v"(xi));
    return __builtin_bit_cast(h2, r);
}
__device__ __forceinline__ h2 exp2_neg_abs_h2(h2 x) {   // 2^-|x| with the source modifiers doing -|.|
    unsigned r;
    const unsigned xi = bits(x);
    asm("v_exp_f16_sdwa %0, -|%1| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 0\n\t"
        "v_exp_f16_sdwa %0, -|%1| dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
        : "=&v"(r) : "v"(xi));
    return __builtin_bit_cast(h2, r);
}
__device__ __forceinline__ h2 log2_h2(h2 x) {
    unsigned r;
    const unsigned xi = bits(x);
    asm("v_log_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 0\n\t"
        "v_log_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"
        : "=&v"(r) : "v"(xi));
    return __builtin_bit_cast(h2, r);
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
// ReLU on the BIT PATTERN: a negative half is a negative int16, so one v_pk_max_i16 against 0 does it (-0 -> +0), without
// the canonicalising extra v_pk_max the float max builtin carries and without inline asm (which the scheduler cannot
// move).
__device__ __forceinline__ h2 relu_h2(h2 z) {
    return __builtin_bit_cast(h2, __builtin_elementwise_max(__builtin_bit_cast(s16x2, z), (s16x2){0, 0}));
}
__device__ __forceinline__ h2 softplus2(h2 z) {   // h' = max(z',0) + log2(1 + 2^-|z'|)
#ifdef MP_EXP_NOTRANS
    return relu_h2(z) + z * (h2){(op_t)0.001f, (op_t)0.001f};
#else
    const h2 u = exp2_neg_abs_h2(z);
#ifdef MP_EXP_LOGTRANS
    return relu_h2(z) + log2_h2(u + (h2){(op_t)1.0f, (op_t)1.0f});
#else   // the same three fused multiply-adds as the phase-separated stream's V program (pp_instr): identical results
    const h2 c1 = {(op_t)1.42459527f, (op_t)1.42459527f}, c2 = {(op_t)-0.58921265f, (op_t)-0.58921265f},
             c3 = {(op_t)0.16538905f, (op_t)0.16538905f};
    h2 t = __builtin_elementwise_fma(u, c3, c2);
    t = __builtin_elementwise_fma(t, u, c1);
    return __builtin_elementwise_fma(t, u, relu_h2(z));
#endif
#endif
}
// lanes 8..15 of every 16-lane row receive lane-8's register, lanes 0..7 keep their own (DPP row_shr:8).  Inline asm on
// purpose (hipcc 7.2 merges two __builtin_amdgcn_update_dpp calls on the elements of a vector into one broadcast);
// s_nop 1 = the VALU-write -> DPP-read hazard.
__device__ __forceinline__ h2 row_shr8(h2 s) {
    unsigned d;
    const unsigned si = bits(s);
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(si), "0"(si));
    return __builtin_bit_cast(h2, d);
}

// Piece q (0..7) of the activation of a finished block of 16 rows x NB column blocks; one piece rides in every K step
// of the next block's MFMA stream.  HIDDEN (compile time: the layer loop dispatches on it, so that the K-step stream
// is straight-line code): apply the nonlinearity, else pass through.
//   plain  : q = 0..3 -> column block q/2, row pair q%2
//   forward: half-block tangent layout (8 points per wave): block 0 = [values | d/dx], block 1 = [d/dy | d/dz]; lanes
//            with (lane & 8) == 0 hold the value / d/dy columns of point lane&7, the others d/dx / d/dz.
//            q = 0, 1: row pair q of both blocks (softplus + sigmoid on the values, tangents scaled by the sigmoid)
template <int NB, bool FWD, int HID, bool HIDDEN, int q, typename NB_T>
__device__ __forceinline__ void act_piece(const f32x4 (&p)[NB], NB_T& Bn, int pc, int ph, u32x4 (&sg)[NB],
                                          const SigIO& sig, int sig_layer) {
    static_assert(NB == 2, "the MLP core is specialised for 2 column blocks per wave (2 waves per SIMD)");
    if constexpr (FWD) {
        if constexpr (q < 2) {
            h2 z = to_h2(p[0][2 * q], p[0][2 * q + 1]);
            h2 t = to_h2(p[1][2 * q], p[1][2 * q + 1]);
            if constexpr (HIDDEN) {
                const bool vl = (threadIdx.x & 8) == 0;
                const h2 h = softplus2(z);
#ifdef MP_EXP_NOTRANS
                const h2 s = z * (h2){(op_t)0.01f, (op_t)0.01f};
#else
                const h2 s = exp2_h2(z - h);          // sigmoid(z') = 2^(z' - h'); meaningful in the value lanes
#endif
                const h2 sf = row_shr8(s);            // tangent lanes take it from their point's value lane
                const h2 zt = z * sf;
                z = vl ? h : zt;
                t = t * sf;
            }
            if (pc < KS_REG) {
                Bn.put(pc, 0, ph, q, z);
                Bn.put(pc, 1, ph, q, t);
            }
        }
    } else {
        if constexpr (q < 4) {
            constexpr int nb = q / 2, j = q % 2;
            h2 z = to_h2(p[nb][2 * j], p[nb][2 * j + 1]);
            if constexpr (HIDDEN) {
                if constexpr (HID == HID_SOFTPLUS_SAVE) {
                    const h2 h = softplus2(z);
#ifdef MP_EXP_NOTRANS
                    const unsigned sv = bits(z * (h2){(op_t)0.01f, (op_t)0.01f});
#else
                    const unsigned sv = bits(exp2_h2(z - h));   // sigmoid(z') = 2^(z' - h')
#endif
                    if (ph == 0) { if (j == 0) sg[nb][0] = sv; else sg[nb][1] = sv; }
                    else { if (j == 0) sg[nb][2] = sv; else sg[nb][3] = sv; }
                    z = h;
                    if (ph == 1 && j == 1 && pc < KS_REG)
                        *(u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (pc * NB + nb) * 1024 + (threadIdx.x & 63) * 16) = sg[nb];
                } else if constexpr (HID == HID_SIGMUL) {
                    const unsigned sv = ph == 0 ? (j == 0 ? sg[nb][0] : sg[nb][1]) : (j == 0 ? sg[nb][2] : sg[nb][3]);
                    z = z * __builtin_bit_cast(h2, sv);
                } else {
                    z = HID == HID_SOFTPLUS ? softplus2(z) : relu_h2(z);
                }
            }
            if (pc < KS_REG) Bn.put(pc, nb, ph, j, z);
        }
    }
}

template <int NB, bool FWD, int HID, bool HIDDEN, int q, typename NB_T>
__device__ __forceinline__ void act_from(const f32x4 (&p)[NB], NB_T& Bn, int pc, int ph, u32x4 (&sg)[NB],
                                         const SigIO& sig, int sig_layer) {
    act_piece<NB, FWD, HID, HIDDEN, q>(p, Bn, pc, ph, sg, sig, sig_layer);
    if constexpr (q + 1 < 8) act_from<NB, FWD, HID, HIDDEN, q + 1>(p, Bn, pc, ph, sg, sig, sig_layer);
}

// Sigmoid fragment buffers: chunk pc's four dwords per column block live in sgb[pc & 1]: written (HID_SOFTPLUS_SAVE) or
// read (HID_SIGMUL, loaded at the chunk's start) from block (pc,1) to block (pc+1,0).  [Loading the reverse sweep's
// fragments one chunk ahead into a third buffer, with or without letting them stay in flight across the chunk
// barrier (s_waitcnt vmcnt(4) instead of 0), was measured 5-7 % SLOWER: the sweep is not waiting for HBM.]
constexpr int SIG_BUFS = 2;
template <int HID>
__device__ __forceinline__ constexpr int sig_slot(int pc) { return pc & 1; }

// Activation of a layer's LAST block (chunk pc_rt = n_chunk - 1, known only at run time, second half block): the one
// piece of the pipeline that has no next block to ride in.  Written with selects on pc_rt -- when it was a copy of
// act_from per unrolled chunk, hipcc merged the copies into one block that indexes the operand array dynamically and
// moved the array to scratch memory.
template <int NB, bool FWD, int HID, bool HIDDEN, typename NB_T>
__device__ __forceinline__ void act_drain(const f32x4 (&p)[NB], NB_T& Bn, int pc_rt, u32x4 (&sgb)[SIG_BUFS][NB],
                                          const SigIO& sig, int sig_layer) {
    const bool odd = pc_rt & 1;
#define MP_SG(nb, i) (odd ? sgb[1][nb][i] : sgb[0][nb][i])
    if constexpr (FWD) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            h2 z = to_h2(p[0][2 * q], p[0][2 * q + 1]);
            h2 t = to_h2(p[1][2 * q], p[1][2 * q + 1]);
            if constexpr (HIDDEN) {
                const bool vl = (threadIdx.x & 8) == 0;
                const h2 h = softplus2(z);
                const h2 s = exp2_h2(z - h);
                const h2 sf = row_shr8(s);
                const h2 zt = z * sf;
                z = vl ? h : zt;
                t = t * sf;
            }
            Bn.put_sel(pc_rt, 0, 1, q, z);
            Bn.put_sel(pc_rt, 1, 1, q, t);
        }
    } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            // the four dwords of this chunk's sigmoid fragment, selected by buffer one by one (explicit scalars: with
            // a u32x4 temporary indexed by the unrolled row-pair counter, hipcc 7.2 multiplied both row pairs by dword 0)
            const unsigned s0 = MP_SG(nb, 0), s1 = MP_SG(nb, 1);
            unsigned s2 = MP_SG(nb, 2), s3 = MP_SG(nb, 3);
            h2 z0 = to_h2(p[nb][0], p[nb][1]), z1 = to_h2(p[nb][2], p[nb][3]);
            if constexpr (HIDDEN) {
                if constexpr (HID == HID_SOFTPLUS_SAVE) {
                    const h2 h0 = softplus2(z0), h1 = softplus2(z1);
                    s2 = bits(exp2_h2(z0 - h0));
                    s3 = bits(exp2_h2(z1 - h1));
                    z0 = h0;
                    z1 = h1;
                } else if constexpr (HID == HID_SIGMUL) {
                    z0 = z0 * __builtin_bit_cast(h2, s2);
                    z1 = z1 * __builtin_bit_cast(h2, s3);
                } else {
                    z0 = HID == HID_SOFTPLUS ? softplus2(z0) : relu_h2(z0);
                    z1 = HID == HID_SOFTPLUS ? softplus2(z1) : relu_h2(z1);
                }
            }
            Bn.put_sel(pc_rt, nb, 1, 0, z0);
            Bn.put_sel(pc_rt, nb, 1, 1, z1);
            if constexpr (HIDDEN && HID == HID_SOFTPLUS_SAVE) {
                if (pc_rt < KS_REG)
                    *(u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (pc_rt * NB + nb) * 1024 + (threadIdx.x & 63) * 16) =
                        (u32x4){s0, s1, s2, s3};
            }
        }
    }
#undef MP_SG
}

constexpr int A_PF = 3, A_QN = 4;   // A-tile prefetch distance / rotating queue length (tiles), interleaved stream
#ifndef MP_PP_PF
#define MP_PP_PF 3
#endif
constexpr int PP_PF = MP_PP_PF, PP_QN = PP_PF + 1;   // the same for the phase-separated stream (its M phase is dense)
constexpr int AQ_LEN = PP_QN > A_QN ? PP_QN : A_QN;
// ---------------------------------------------------------------------------------------------------------------------
// Phase-separated ("ping-pong") hidden softplus layers, plain mode.
//
// Measured on MI355X (tools/valu_rate.hip, tools/issue_model.hip): packed-half VALU instructions cost 4 cycles of the
// SIMD's vector pipe, v_exp/v_log 8, a wave issues at most one instruction per ~5 cycles, and MFMA 16x16x32 occupies
// the matrix pipe for 16.  A softplus(+sigmoid) layer needs 52 (72) VALU-pipe cycles per row pair against 64 matrix-pipe
// cycles: both pipes are needed almost all the time, so they must run CONCURRENTLY.  When both waves of a SIMD run the
// same finely interleaved MFMA + activation stream they do not: the stream of act_piece keeps the matrix pipe ~30 % busy.
// Here a wave's stream per weight chunk is two homogeneous phases,
//     M(c): the chunk's 32 MFMAs back to back (4 independent accumulators: 2 row blocks x 2 column blocks), then
//     V(c): the activation of its 8 row pairs in lock step (one instruction per pair and stage: every producer is 8
//           instructions back -- no s_nop for the transcendental / SDWA result hazards, no dependent-issue stall),
// and the two waves of a SIMD run them in ANTI-PHASE: waves 0..3 (one per SIMD) put the chunk barrier after V(c), waves
// 4..7 between M(c) and V(c).  Between two barriers a SIMD therefore sees  [M(c) V(c)]  from one wave and
// [V(c-1) M(c)]  from the other: one wave's MFMAs beside the other's VALU work.  The ring protocol is unchanged (every
// wave passes one barrier per chunk, after its M(c): chunk c's slot may be refilled, chunk c+1 is complete), and V(c)
// has a compile-time chunk index, so the run-time "drain" of the layer's last block does not exist on this path.
// Arithmetic per element is exactly act_piece's.
struct ActRegs8 {
    unsigned z[8], u[8], lg[8], r[8], h[8], d[8], s[8];
};
// log2(1 + u) on u in [0, 1] as u (C1 + u (C2 + u C3)): minimax cubic (max error 7.7e-4; evaluated in packed half 1.3e-3,
// the v_exp / +1 / v_log chain it replaces loses u below 2^-11 in the half-precision sum 1 + u and is no better).  Three
// packed FMAs instead of an addition and two transcendentals per row pair -- beside the partner wave's MFMAs a v_log costs
// 13.3 cycles of the SIMD's vector pipe, a packed FMA 8.25 (tools/pair_model.hip) -- and the last FMA adds max(z', 0).
#ifdef MP_EXP_LOGTRANS   // ablation: the transcendental chain
constexpr bool LOG_POLY = false;
#else
constexpr bool LOG_POLY = true;
#endif
constexpr float LOG2P_C1 = 1.42459527f, LOG2P_C2 = -0.58921265f, LOG2P_C3 = 0.16538905f;
// The stored sigmoids travel through HBM as UNORM8 (default; -DMP_EXP_SIG16: as halves, 4 KiB per point).  At half precision the
// forward sweep writes 512 B per point and hidden layer: at a fraction f of the MFMA peak that is f x 9.8 TB/s of stores (the
// reverse sweep: the same in loads) -- HBM caps the forward sweep at ~0.45 of the peak and the reverse sweep at ~0.54 whatever
// the instruction stream does (tools/stream_model.hip reproduces it).  A byte per sigmoid halves both.
//   quantise (forward):  t = 255 s + 1024 in half precision (one v_pk_fma_f16, round to nearest): the low byte of a half in
//                        [1024, 2048) IS its integer part - 1024 = round(255 s); the four low bytes of two row pairs are gathered
//                        into one dword by one v_perm_b32.   +1.5 VALU per row pair, 8 B instead of 16 per lane, chunk and block
//   restore (reverse):   v_perm_b32 spreads two bytes under 0x64 (= the halves 1024 + b), - 1024, x 1/255.   +3 VALU per row pair
// sigmoid(100 z) of the beta = 100 softplus is 0 or 1 to within a step for all but the units in transition: the error is
// <= 1/510 absolute on those and ZERO on the saturated ones (0 -> 0, 1 -> 255 -> 1).
#ifdef MP_EXP_SIG16
constexpr bool SIG8 = false;
#else
constexpr bool SIG8 = true;
#endif
// -DMP_EXP_SIGBITS=4 | 6: the PRECISION of a 4- / 6-bit uniform code (levels k / 15, k / 63: exact 0 and 1 + 14 / 62 interior
// levels) emulated inside the byte pipeline -- the same instructions, only the two scale constants change, so the normals of
// such a code can be measured (tests/test_mlp_gpu.py, tools/sig_bits.py) before any packing code is written.  Round-5 result:
// DESIGN.md section 3 -- the 4-bit code does not hold the asserted normal bounds.
#ifndef MP_EXP_SIGBITS
#define MP_EXP_SIGBITS 8
#endif
constexpr unsigned SIG_Q = MP_EXP_SIGBITS == 4 ? 0x4B804B80u : MP_EXP_SIGBITS == 6 ? 0x53E053E0u : 0x5BF85BF8u;      // 15.0 | 63.0 | 255.0
constexpr unsigned SIG_QINV = MP_EXP_SIGBITS == 4 ? 0x2C442C44u : MP_EXP_SIGBITS == 6 ? 0x24102410u : 0x1C041C04u;   // 1/15 | 1/63 | 1/255
constexpr unsigned SIG_Q_INT = MP_EXP_SIGBITS == 4 ? 15u : MP_EXP_SIGBITS == 6 ? 63u : 255u;
constexpr float SIG_QINV_F = MP_EXP_SIGBITS == 4 ? 1.0f / 15.0f : MP_EXP_SIGBITS == 6 ? 1.0f / 63.0f : 1.0f / 255.0f;
constexpr int SIG_CHUNK_BYTES = SIG8 ? 1024 : 2048;   // per wave and chunk: 64 lanes x (2 column blocks x 4 row pairs x 2 sigmoids)
// -DMP_SIG_FROM_H (round 5): the stored sigmoid as 1 - 2^(-h') instead of 2^(z' - h')  [2^h' = 1 + 2^z', so both equal
// 2^z' / (1 + 2^z')]: the subtraction z' - h' disappears (the negation rides on the v_exp source modifier) and the "1 -" folds
// into the byte quantisation's FMA: t = -255 e + 1279.  One VALU stage fewer per row pair in the forward sweep (11.5 instead
// of 12.5), which is bound by exactly that program.  The difference 1 - e loses RELATIVE precision where the sigmoid is tiny
// (absolute error <= 2^-11 from the rounding of e near 1), far below the byte's 1 / 510.
#ifdef MP_SIG_FROM_H
constexpr bool SIG_FROM_H = SIG8;
#else
constexpr bool SIG_FROM_H = false;
#endif
struct ActConst {
    unsigned c1, c2, c3;   // the coefficients as packed half pairs, in vector registers (gfx9 VOP3P: no literals, one SGPR)
    unsigned c1024, c64;   // 8-bit sigmoids: 1024.0 pairs; 0x64646464 (the high bytes of halves 1024 + b)
    unsigned ctop;         // SIG_FROM_H: (1024 + Q).0 pairs, Q = 255 (the byte of sigmoid 1)
};
__device__ __forceinline__ ActConst act_const() {
    const h2 a = {(op_t)LOG2P_C1, (op_t)LOG2P_C1}, b = {(op_t)LOG2P_C2, (op_t)LOG2P_C2}, c = {(op_t)LOG2P_C3, (op_t)LOG2P_C3};
    const h2 k = {(op_t)1024.0f, (op_t)1024.0f};
    return ActConst{bits(a), bits(b), bits(c), bits(k), 0x64646464u, 0x64006400u + (SIG_Q_INT | (SIG_Q_INT << 16))};
}
__device__ __forceinline__ unsigned sconst(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }   // into an SGPR
// stages of the V program per layer kind (HIDDEN = false: the linear output layers, conversion only)
__host__ __device__ constexpr int pp_stages(int hid, bool hidden) {
    return !hidden ? 1 : hid == HID_SOFTPLUS_SAVE ? (LOG_POLY ? 10 : 11) + (SIG8 ? 2 : 0) - (SIG_FROM_H ? 1 : 0) : hid == HID_SOFTPLUS ? (LOG_POLY ? 7 : 8)
                     : hid == HID_SIGMUL && SIG8 ? 5 : 2;
}

// stage ST for row pair q = 4 mbl + 2 nb + j  (mbl: 16-row block of the chunk, nb: column block, j: row pair)
template <int HID, bool HIDDEN, int ST, int q, typename NB_T>
__device__ __forceinline__ void pp_instr(ActRegs8& a, const ActConst& k, const f32x4 (&acc)[CHUNK_MB][2], NB_T& Bn, int c,
                                         u32x4 (&sg)[2]) {
    constexpr int mbl = q / 4, nb = (q / 2) % 2, j = q % 2;
    constexpr bool SP = HIDDEN && (HID == HID_SOFTPLUS || HID == HID_SOFTPLUS_SAVE);
    // softplus stage ids: E0 E1 = u = 2^-|z'| (low half writes the dword, high half in place); then either
    //   polynomial: P0 t = C3 u + C2, P1 t = t u + C1, RL r = max(z', 0), HH h' = t u + r
    //   transcendental: A1 u += 1, L0 L1 lg = log2(u), RL, HH h' = r + lg
    // and for the stored sigmoids: SB d = z' - h', S0 S1 sigmoid = 2^d
    constexpr int E0 = 1, E1 = 2, P0 = LOG_POLY ? 3 : -1, P1 = LOG_POLY ? 4 : -1, A1 = LOG_POLY ? -1 : 3, L0 = LOG_POLY ? -1 : 4,
                  L1 = LOG_POLY ? -1 : 5, RL = LOG_POLY ? 5 : 6, HH = LOG_POLY ? 6 : 7, SB = SIG_FROM_H ? -1 : HH + 1,
                  S0 = SIG_FROM_H ? HH + 1 : HH + 2, S1 = S0 + 1;
    if constexpr (ST == 0) {          // left to the compiler: it knows the MFMA -> VALU read hazard
        a.z[q] = bits(to_h2(acc[mbl][nb][2 * j], acc[mbl][nb][2 * j + 1]));
        if constexpr (!HIDDEN) {
            if (c < KS_REG) Bn.put(c, nb, mbl, j, __builtin_bit_cast(h2, a.z[q]));
        }
    } else if constexpr (HID == HID_SIGMUL && SIG8) {   // restore the byte-sized sigmoids, then multiply
        static_assert(ST >= 1 && ST <= 4 && HIDDEN, "pp_instr: stage");
        // sg[0] = this chunk's 16 bytes: dword 2 nb + mbl holds row pairs j = 0, 1 of block (mbl, nb) as bytes (2 j, 2 j + 1)
        constexpr int e = 2 * nb + mbl;
        if constexpr (ST == 1) {
            unsigned src;
            if constexpr (e == 0) src = sg[0].x;
            else if constexpr (e == 1) src = sg[0].y;
            else if constexpr (e == 2) src = sg[0].z;
            else src = sg[0].w;
            // D = bytes of {S0 = 0x64.., S1 = src}: (src byte 2 j, 0x64, src byte 2 j + 1, 0x64) = the halves 1024 + b
            asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a.u[q]) : "v"(k.c64), "v"(src), "s"(sconst(j == 0 ? 0x04010400u : 0x04030402u)));
        } else if constexpr (ST == 2) {
            asm volatile("v_pk_add_f16 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a.u[q]) : "v"(k.c1024));
        } else if constexpr (ST == 3) {
            asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a.u[q]) : "s"(sconst(SIG_QINV)));     // x 1/255 (0x1C04 = 0.0039215)
        } else {
            const h2 z = __builtin_bit_cast(h2, a.z[q]) * __builtin_bit_cast(h2, a.u[q]);
            if (c < KS_REG) Bn.put(c, nb, mbl, j, z);
        }
    } else if constexpr (!SP) {       // ReLU / multiplication by the stored sigmoid: one packed instruction
        static_assert(ST == 1 && HIDDEN, "pp_instr: stage");
        h2 z = __builtin_bit_cast(h2, a.z[q]);
        if constexpr (HID == HID_SIGMUL) {
            // dword 2 mbl + j of the fragment, picked by NAME: hipcc 7.2 folds a subscript on the loaded u32x4 to
            // element 0 here (same miscompile as noted at act_drain)
            constexpr int e = 2 * mbl + j;
            unsigned sv;
            if constexpr (e == 0) sv = sg[nb].x;
            else if constexpr (e == 1) sv = sg[nb].y;
            else if constexpr (e == 2) sv = sg[nb].z;
            else sv = sg[nb].w;
            z = z * __builtin_bit_cast(h2, sv);
        } else {
            z = relu_h2(z);
        }
        if (c < KS_REG) Bn.put(c, nb, mbl, j, z);
    } else if constexpr (ST == E0) {
        asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(a.u[q]) : "v"(a.z[q]));
    } else if constexpr (ST == E1) {
        asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a.u[q]) : "v"(a.z[q]));
    } else if constexpr (ST == P0) {
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a.lg[q]) : "v"(a.u[q]), "v"(k.c3), "v"(k.c2));
    } else if constexpr (ST == P1) {
        asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a.lg[q]) : "v"(a.u[q]), "v"(k.c1));
    } else if constexpr (ST == A1) {
        asm volatile("v_pk_add_f16 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(a.u[q]));
    } else if constexpr (ST == L0) {
        asm volatile("v_log_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(a.lg[q]) : "v"(a.u[q]));
    } else if constexpr (ST == L1) {
        asm volatile("v_log_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a.lg[q]) : "v"(a.u[q]));
    } else if constexpr (ST == RL) {  // max(z', 0) on the bit pattern
        asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(a.r[q]) : "v"(a.z[q]));
    } else if constexpr (ST == HH) {  // h' = max(z', 0) + log2(1 + 2^-|z'|): the next layer's operand register as it stands
        if constexpr (LOG_POLY) asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(a.h[q]) : "v"(a.lg[q]), "v"(a.u[q]), "v"(a.r[q]));
        else asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a.h[q]) : "v"(a.r[q]), "v"(a.lg[q]));
        if (c < KS_REG) Bn.put(c, nb, mbl, j, __builtin_bit_cast(h2, a.h[q]));
    } else if constexpr (ST == SB) {  // z' - h'
        asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a.d[q]) : "v"(a.z[q]), "v"(a.h[q]));
    } else if constexpr (ST == S0) {  // sigmoid(z') = 2^(z' - h');  SIG_FROM_H: e = 2^(-h') = 1 - sigmoid(z')
        if constexpr (SIG_FROM_H)
            asm volatile("v_exp_f16_sdwa %0, -%1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(a.s[q]) : "v"(a.h[q]));
        else
            asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(a.s[q]) : "v"(a.d[q]));
    } else if constexpr (ST == S1) {
        if constexpr (SIG_FROM_H)
            asm volatile("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a.s[q]) : "v"(a.h[q]));
        else
            asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a.s[q]) : "v"(a.d[q]));
        if constexpr (!SIG8) {
            constexpr int e = 2 * mbl + j;   // by name, see the note at the HID_SIGMUL read
            if constexpr (e == 0) sg[nb].x = a.s[q];
            else if constexpr (e == 1) sg[nb].y = a.s[q];
            else if constexpr (e == 2) sg[nb].z = a.s[q];
            else sg[nb].w = a.s[q];
        }
    } else if constexpr (ST == S1 + 1) {   // 8-bit: 1024 + 255 s (0x5BF8 = 255.0): the mantissa's low byte is round(255 s)
        static_assert(SIG8, "pp_instr: stage");
        if constexpr (SIG_FROM_H)     // 1024 + Q (1 - e) = -Q e + (1024 + Q)
            asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a.s[q]) : "s"(sconst(SIG_Q | 0x80008000u)), "v"(k.ctop));
        else
            asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a.s[q]) : "s"(sconst(SIG_Q)), "v"(k.c1024));
    } else {                               // 8-bit: the low bytes of the four halves of row pairs j = 0, 1 into one dword
        static_assert(SIG8 && ST == S1 + 2, "pp_instr: stage");
        if constexpr (j == 0) {
            unsigned d;
            // D = bytes of {S0 = pair j = 1, S1 = pair j = 0}: S1.b0, S1.b2, S0.b0, S0.b2
            asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a.s[q + 1]), "v"(a.s[q]), "s"(sconst(0x06040200u)));
            constexpr int e = 2 * nb + mbl;
            if constexpr (e == 0) sg[0].x = d;
            else if constexpr (e == 1) sg[0].y = d;
            else if constexpr (e == 2) sg[0].z = d;
            else sg[0].w = d;
        }
    }
}
template <int HID, bool HIDDEN, int I, typename NB_T>
__device__ __forceinline__ void pp_program(ActRegs8& a, const ActConst& k, const f32x4 (&acc)[CHUNK_MB][2], NB_T& Bn, int c,
                                           u32x4 (&sg)[2]) {
    if constexpr (I < 8 * pp_stages(HID, HIDDEN)) {
        pp_instr<HID, HIDDEN, I / 8, I % 8>(a, k, acc, Bn, c, sg);
        pp_program<HID, HIDDEN, I + 1>(a, k, acc, Bn, c, sg);
    }
}

// V phase of HID_SOFTPLUS_X2: fp32 softplus (scaled units, base 2) of the chunk's 8 rows per lane, z' = (W_h x_h) + (W_h x_l), and the
// split of the result into the next layer's two operand blocks: hi = half(h'), lo = half(h' - hi).  Plain C++ between the two
// scheduling barriers of the V phase: eight independent chains, the compiler interleaves them and knows the transcendental hazards.
template <typename NB_T>
__device__ __forceinline__ void x2_program(const f32x4 (&acc)[CHUNK_MB][2], NB_T& Bn, int c) {
    float h[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float z = acc[q >> 2][0][q & 3] + acc[q >> 2][1][q & 3];
        const float u = __builtin_amdgcn_exp2f(-__builtin_fabsf(z));
        h[q] = __builtin_fmaxf(z, 0.0f) + __builtin_amdgcn_logf(1.0f + u);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {          // row pair j = p & 1 of 16-row block mbl = p >> 1
        const h2 hi = to_h2(h[2 * p], h[2 * p + 1]);
        const h2 lo = to_h2(h[2 * p] - (float)hi[0], h[2 * p + 1] - (float)hi[1]);
        if (c < KS_REG) {
            Bn.put(c, 0, p >> 1, p & 1, hi);
            Bn.put(c, 1, p >> 1, p & 1, lo);
        }
    }
}

// Which weight chunks have register-fed / input-fed K steps (a chunk's unused tiles are never fetched): bit ci of the
// two masks, built once per network from the layer table.
struct ChunkMasks {
    unsigned long long reg[2], in[2];
    __device__ __forceinline__ bool has_reg(int ci) const { return ((ci < 64 ? reg[0] >> ci : reg[1] >> (ci - 64)) & 1ull) != 0; }
    __device__ __forceinline__ bool has_in(int ci) const { return ((ci < 64 ? in[0] >> ci : in[1] >> (ci - 64)) & 1ull) != 0; }
};
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {   // keeps the value in scalar registers
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ ChunkMasks chunk_masks(const NetDesc& net) {
    unsigned long long r0 = 0, r1 = 0, i0 = 0, i1 = 0;
    int ci = 0;
    for (int l = 0; l < net.n_layers; ++l) {
        const int n = net.layer[l].n_chunk;
        // n consecutive bits from bit ci (n <= 9): split over the two words
        const unsigned long long run = (1ull << n) - 1ull;
        const unsigned long long lo = ci < 64 ? run << ci : 0ull;
        const unsigned long long hi = ci >= 64 ? run << (ci - 64) : (ci + n > 64 ? run >> (64 - ci) : 0ull);
        if (net.layer[l].use_reg) { r0 |= lo; r1 |= hi; }
        if (net.layer[l].use_in) { i0 |= lo; i1 |= hi; }
        ci += n;
    }
    return ChunkMasks{{uniform_u64(r0), uniform_u64(r1)}, {uniform_u64(i0), uniform_u64(i1)}};
}

// The phase-separated stream's weight DMA: issued by the LATE waves only (wl = wave - WAVES/2 = 0..3), right behind
// their barrier, i.e. at the head of a V phase -- an LDS-DMA instruction costs 25-60 cycles to issue among VALU work
// and 100-185 among MFMAs (MI355X_MICROARCH.md), and it is waited for a whole M phase later.  Pieces of chunk `cn` (1 KiB
// each, chunk layout [row block][8 register-fed tiles | KS_IN input-fed tiles]): wave wl takes register-fed tiles
// {wl, wl + 4} of both row blocks and input-fed pieces {wl, wl + 4} of the 2 KS_IN; tiles the chunk's layer never
// multiplies are skipped.  Addressing: ONE scalar source base per chunk + per-piece lane offsets prepared once per
// network (DmaLanes), the LDS destination through M0 (s_add per piece).
template <int KS_IN>
struct DmaLanes {
    unsigned reg[4];                 // lane*16 + byte offset of this wave's four register-fed pieces
    unsigned in[(2 * KS_IN + 3) / 4 + 1];   // ... of its input-fed pieces (the last may not exist; + 1: never a zero-length array)
    unsigned run;                    // MP_DMA_ONE_M0: lane*16 + byte offset of this wave's contiguous 4 KiB run of register-fed tiles
    unsigned wl;
};
template <int KS_IN>
__device__ __forceinline__ DmaLanes<KS_IN> dma_lanes(int wl, int lane) {
    DmaLanes<KS_IN> d;
    d.wl = wl;
    d.run = lane * 16 + (wl >> 1) * mb_bytes(KS_IN) + (wl & 1) * 4 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) d.reg[i] = lane * 16 + (i / 2) * mb_bytes(KS_IN) + (wl + 4 * (i % 2)) * TILE_BYTES;
    if constexpr (KS_IN > 0) {
#pragma unroll
        for (int i = 0; i < (2 * KS_IN + 3) / 4; ++i) {
            const int k = wl + 4 * i;                      // input-fed piece index: row block k / KS_IN, tile k % KS_IN
            d.in[i] = lane * 16 + (k / KS_IN) * mb_bytes(KS_IN) + (KS_REG + k % KS_IN) * TILE_BYTES;
        }
    }
    return d;
}
__device__ __forceinline__ void pp_dma_piece(const char* src_chunk_uniform, unsigned lane_off, unsigned lds_chunk_base,
                                             unsigned piece_off_uniform) {
#ifdef MP_EXP_DMA1LANE   // ablation (timing only, results are wrong): the same instructions, ONE lane's 16 bytes instead of 1 KiB per piece
    asm volatile("s_mov_b64 s[98:99], exec\n\ts_mov_b64 exec, 1\n\ts_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b64 exec, s[98:99]"
                 ::"s"(lds_chunk_base), "s"(piece_off_uniform), "v"(lane_off), "s"(src_chunk_uniform) : "memory", "s98", "s99");
#else
    asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                 ::"s"(lds_chunk_base), "s"(piece_off_uniform), "v"(lane_off), "s"(src_chunk_uniform) : "memory");
#endif
}
// PART: 0 = all of this wave's pieces, 1 = those of row block 0 only, 2 = those of row block 1 only (-DMP_DMA_SPLIT)
template <int KS_IN, int PART = 0>
__device__ __forceinline__ void pp_issue(const char* __restrict__ wpack, char* wring, int cn, int ring_slot,
                                         const ChunkMasks& cm, const DmaLanes<KS_IN>& d) {
    constexpr int CB = chunk_bytes(KS_IN);
    const int cn_u = __builtin_amdgcn_readfirstlane(cn);
    const char* src = uniform_ptr(wpack) + (size_t)cn_u * CB;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_offset(wring)) + __builtin_amdgcn_readfirstlane(ring_slot) * CB;
    const unsigned wl = __builtin_amdgcn_readfirstlane(d.wl);
    if (cm.has_reg(cn_u)) {
#ifdef MP_DMA_ONE_M0
        // ONE M0 write per chunk: wave wl takes the register-fed tiles 4 (wl & 1) .. + 3 of row block wl >> 1 -- a contiguous
        // 4 KiB run in the chunk (source and ring slot have the same layout), its four pieces addressed through the
        // instruction's immediate offset, which moves the global AND the LDS address (tools/stream_model.hip: +2..5 % on
        // the forward sweep, +1..4 % on the reverse sweep against one s_add + s_nop per piece)
        const unsigned run = (wl >> 1) * mb_bytes(KS_IN) + (wl & 1) * 4 * TILE_BYTES;
        asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\tglobal_load_lds_dwordx4 %2, %3 offset:1024\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:2048\n\tglobal_load_lds_dwordx4 %2, %3 offset:3072"
                     ::"s"(dst), "s"(run), "v"(d.run), "s"(src) : "memory");
#else
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (PART == 0 || i / 2 == PART - 1)
                pp_dma_piece(src, d.reg[i], dst, (i / 2) * mb_bytes(KS_IN) + (wl + 4 * (i % 2)) * TILE_BYTES);
#endif
    }
    if constexpr (KS_IN > 0) {
        if (cm.has_in(cn_u)) {
#pragma unroll
            for (int i = 0; i < (2 * KS_IN + 3) / 4; ++i) {
                const unsigned k = wl + 4 * i;
                if (k < 2 * KS_IN && (PART == 0 || k / KS_IN == PART - 1))
                    pp_dma_piece(src, d.in[i], dst, (k / KS_IN) * mb_bytes(KS_IN) + (KS_REG + k % KS_IN) * TILE_BYTES);
            }
        }
    }
}

struct NoCapture {
    template <int NB>
    __device__ __forceinline__ void operator()(int, int, const f32x4 (&)[NB]) const {}
};

// One layer of the phase-separated stream (every layer kind of the plain-mode kernels: softplus with or without the
// stored sigmoids, ReLU, the reverse sweep's multiplication by the stored sigmoids, and -- HIDDEN = false -- the linear
// output layers).  `pp_last`: last chunk index of the network.
template <int KS_IN, int HID, int WAVES, bool HIDDEN, typename Cap, typename NB_T>
__device__ __forceinline__ void run_layer_pp(const NetDesc& net, const LayerDesc L, int l, const char* __restrict__ wpack,
                                             const float* bias_lds, char* wring, opx8 (&Bcur)[KS_REG][2], NB_T& Bn,
                                             opx8 (&aq)[AQ_LEN], u32x4 (&sgb)[SIG_BUFS][2], const op_t* stage_wave,
                                             f32x4 (&out)[2], int wave, int lane, const SigIO& sig, Cap& cap, int& ci,
                                             int& ring_pos, const ChunkMasks& cm, const DmaLanes<KS_IN>& dl, int pp_last) {
    constexpr int PF = PP_PF, QN = PP_QN, NT = 2 * KS_REG;   // register-fed A tiles of a chunk, order (ks, mbl)
    constexpr bool BIAS = HID != HID_SIGMUL;                 // the reverse sweep has no bias
    const int g = lane >> 4;
    const float* bl = BIAS ? bias_lds + l * BIAS_STRIDE + g * 4 : nullptr;
    const int sig_layer = HID == HID_SIGMUL ? (L.aux & 0xff) - 1 : l;
    const int cap_id = (L.aux >> 8) & 0xff;
    // second wave of its SIMD (a workgroup's waves are dealt to the 4 SIMDs cyclically): barrier between M(c) and V(c).
    // readfirstlane: a scalar branch, not an exec-masked region
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= WAVES / 2;
    const ActConst kact = act_const();
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
        if ((HIDDEN ? c < KS_REG : true) && c < L.n_chunk) {
            MP_STAMP(0);
            // ring_pos = ci % RING_SLOTS, carried along (a division by 3 per use costs ~8 scalar instructions)
            const int ring_next = ring_pos + 1 == RING_SLOTS ? 0 : ring_pos + 1;
            const char* slot = wring + ring_pos * chunk_bytes(KS_IN) + lane * 16;
            const char* slot_next = wring + ring_next * chunk_bytes(KS_IN) + lane * 16;
            // Reverse sweep: the stored sigmoids are fetched ONE CHUNK AHEAD within a layer (chunk c + 1's fragment is
            // requested at the head of chunk c and renamed into place behind V(c); a layer's first fragment is requested
            // at its own head): an HBM read under this load takes longer than an M phase, and a fragment fetched at the
            // head of its own chunk stalled every V phase; the wait behind M(c) is COUNTED -- vmcnt(2), everything but the
            // loads just issued.  [tools/stream_model.hip: 0.45 -> 0.55 of the MFMA peak for this stream.  Carrying
            // the fragment across the LAYER loop as well costs hipcc 7.2 the register file: 256 VGPRs + 112 spilled, from
            // 208 -- and scratch traffic counts in vmcnt.]
            constexpr bool REV = HID == HID_SIGMUL && HIDDEN;
#ifdef MP_EXP_NOAHEAD   // ablation: every fragment at the head of its own chunk, full drain behind M(c)
            constexpr bool AHEAD = false;
#else
            constexpr bool AHEAD = REV;
#endif
            u32x4 (&sg)[2] = sgb[0];
            u32x4 sgn[2];
            constexpr int SGV = SIG8 ? 1 : 2;   // 16-byte vectors per lane and chunk
            if constexpr (REV) {
                if (!AHEAD || c == 0) {
#pragma unroll
                    for (int v = 0; v < SGV; ++v)
                        sg[v] = *(const u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (c * SGV + v) * 1024 + lane * 16);
                }
                if (AHEAD && c + 1 < KS_REG) {
#pragma unroll
                    for (int v = 0; v < SGV; ++v)
                        sgn[v] = *(const u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + ((c + 1) * SGV + v) * 1024 + lane * 16);
                }
            }
            // ---- M(c): 4 accumulators (block mbl, column block nb), biased
            f32x4 acc[CHUNK_MB][2];
#pragma unroll
            for (int mbl = 0; mbl < CHUNK_MB; ++mbl) {
                f32x4 bv = (f32x4){0, 0, 0, 0};
                if constexpr (BIAS) bv = *(const f32x4*)(bl + (2 * c + mbl) * 16);
                acc[mbl][0] = bv;
                acc[mbl][1] = HID == HID_SOFTPLUS_X2 ? (f32x4){0, 0, 0, 0} : bv;   // X2: both blocks are the same points (hi | lo)
            }
            if (L.use_reg) {   // tiles 0 .. PF-1 are in the queue already (loaded at the end of the previous chunk)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int ks = t / 2, mbl = t % 2;
                    const opx8 a = aq[t % QN];
#ifndef MP_EXP_NOLDS
                    if (t + PF < NT) {
                        const int tn = t + PF;
                        aq[tn % QN] = *(const opx8*)(slot + (tn % 2) * mb_bytes(KS_IN) + (tn / 2) * TILE_BYTES);
                    }
#endif
                    // pinned: left alone, the scheduler sinks each read to just before its use and the M phase stalls
                    // on LDS latency at every tile (measured: 850 instead of 500 cycles)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
                        acc[mbl][nb] = MP_MFMA_F16(a, Bcur[ks][nb], acc[mbl][nb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (KS_IN > 0) {
                if (L.use_in) {
#pragma unroll
                    for (int ks = 0; ks < KS_IN; ++ks) {
                        opx8 bi[2];
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            bi[nb] = *(const opx8*)(stage_wave + (nb * 16 + (lane & 15)) * in_stride(KS_IN) + ks * 32 + g * 8);
#pragma unroll
                        for (int mbl = 0; mbl < CHUNK_MB; ++mbl) {
                            const opx8 a = *(const opx8*)(slot + mbl * mb_bytes(KS_IN) + (KS_REG + ks) * TILE_BYTES);
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb)
                                acc[mbl][nb] = MP_MFMA_F16(a, bi[nb], acc[mbl][nb], 0, 0, 0);
                        }
                    }
                }
            }
            if (cap_id == 1) {          // rows 0..47: blocks (0,0), (0,1), (1,0)
                if (c == 0) { cap(1, 0, acc[0]); cap(1, 1, acc[1]); }
                else if (c == 1) cap(1, 2, acc[0]);
            } else if (cap_id == 2) {   // rows 208..255: blocks (6,1), (7,0), (7,1)
                if (c == 6) cap(2, 0, acc[1]);
                else if (c == 7) { cap(2, 1, acc[0]); cap(2, 2, acc[1]); }
            }
            if constexpr (!HIDDEN) {
                if (c == 0 || c == MAX_CHUNKS - 1) {
                    if (c == L.out_chunk) {  // fp32 rows 0..15 of the out chunk
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) out[nb] = acc[0][nb];
                    }
                }
            }
            MP_STAMP(1);
            // The reverse sweep's sigmoid loads (issued before M(c)) are waited for HERE by both kinds of wave: hipcc
            // does not know the asm-issued DMA, so a wait it inserted in V(c) -- reached by the late waves right behind
            // their DMA issue -- would wait for the fresh DMA pieces as well (measured: +9 % on k_mlp_grad).
            // In flight, oldest first: [this chunk's fragment, fetched a chunk ago] [late waves: the DMA pieces issued behind
            // the previous barrier] [the two loads just issued for the next chunk] -- the counted wait covers the first two.
            if constexpr (REV) {
                if (AHEAD && c + 1 < KS_REG) __builtin_amdgcn_s_waitcnt(0x0F70 | SGV);   // vmcnt(SGV): all but the loads just issued
                else dma_wait_all();
            }
            if (late) {
#if defined(MP_DMA_SPLIT)
                // both halves of the workgroup issue: the late waves row block 1 of chunk ci + 3 here, the early waves row block 0 of
                // chunk ci + 2 at the tail of their V phase (below)
                if constexpr (!REV) dma_wait_all();
                __syncthreads();
                if (ci + 3 <= pp_last) pp_issue<KS_IN, 2>(wpack, wring, ci + 3, ring_pos, cm, dl);
#elif defined(MP_DMA_EARLY)
                __syncthreads();   // the late waves issue no DMA in this variant: nothing of theirs to wait for
#else
                // every DMA piece this wave issued (one whole M phase ago) has landed; every wave is done with chunk ci:
                // its ring slot is free for chunk ci + 3
                if constexpr (!REV) dma_wait_all();
                __syncthreads();
#ifndef MP_EXP_NOLOAD
                if (ci + 3 <= pp_last) pp_issue<KS_IN>(wpack, wring, ci + 3, ring_pos, cm, dl);   // (ci + 3) % 3 == ring_pos
#endif
#endif
                MP_STAMP(2);
            }
            // ---- V(c)
            __builtin_amdgcn_sched_barrier(0);
            {
                ActRegs8 a;
#ifdef MP_EXP_NOV   // ablation: no V phase (accumulators kept alive)
                asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
#else
                if constexpr (HID == HID_SOFTPLUS_X2) {
                    if constexpr (HIDDEN) x2_program(acc, Bn, c);   // the linear output layer: `out` holds the two partial sums
                } else {
                    pp_program<HID, HIDDEN, 0>(a, kact, acc, Bn, c, sg);
                }
#endif
                if constexpr (HID == HID_SOFTPLUS_SAVE && HIDDEN) {
#pragma unroll
                    for (int v = 0; v < SGV; ++v)
                        *(u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (c * SGV + v) * 1024 + lane * 16) = sg[v];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (REV) {
                if (AHEAD && c + 1 < KS_REG) {
#pragma unroll
                    for (int v = 0; v < SGV; ++v) sg[v] = sgn[v];
                }
            }
            if (!late) {
                MP_STAMP(2);
#ifdef MP_DMA_EARLY
                // Variant (round 6): the EARLY waves issue the weight DMA, at the tail of their V phase -- where they otherwise
                // wait at the barrier for the late waves, whose [barrier, DMA issue, V, M] chain is the chunk's critical path.
                // Behind the barrier of chunk ci - 1 (passed at the tail of the previous iteration) slot (ci - 1) % 3 is free:
                // chunk ci + 2 goes there; it is first read behind the barrier of chunk ci + 1, before which this wave waits
                // for it (the wait below, one iteration from now).  The forward sweep's sigmoid stores of THIS V phase are the
                // youngest operations in flight: the wait is counted so that it does not include them.
                if constexpr (HID == HID_SOFTPLUS_SAVE && HIDDEN) __builtin_amdgcn_s_waitcnt(0x0F70 | (SIG8 ? 1 : 2));
                else if constexpr (!REV) dma_wait_all();
#ifndef MP_EXP_NOLOAD
#ifdef MP_DMA_SPLIT
                if (ci >= 1 && ci + 2 <= pp_last)
                    pp_issue<KS_IN, 1>(wpack, wring, ci + 2, ring_pos == 0 ? RING_SLOTS - 1 : ring_pos - 1, cm, dl);
#else
                if (ci >= 1 && ci + 2 <= pp_last)
                    pp_issue<KS_IN>(wpack, wring, ci + 2, ring_pos == 0 ? RING_SLOTS - 1 : ring_pos - 1, cm, dl);
#endif
#endif
#endif
                __syncthreads();   // default: the early waves issue no DMA: nothing to wait for (their stores need no wait)
            }
            MP_STAMP(3);
            // the next chunk's first tiles (complete in the ring: both kinds of wave are past the barrier behind M(c))
#pragma unroll
            for (int t = 0; t < PF; ++t)
                aq[t % QN] = *(const opx8*)(slot_next + (t % 2) * mb_bytes(KS_IN) + (t / 2) * TILE_BYTES);
            ++ci;
            ring_pos = ring_next;
        }
    }
#pragma unroll
    for (int k = 0; k < KS_REG; ++k)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) Bcur[k][nb] = Bn.get(k, nb);
}


// K step of the next block that carries the first activation piece of a finished block.  The reverse sweep's pieces
// (one multiplication each) sit in K steps 4..7, away from the MFMAs that produced their inputs: -8 % on k_mlp_grad;
// the softplus pieces are best right at the start (+8 % on k_mlp_fwdsave when shifted).
__device__ __forceinline__ constexpr int act_shift(int hid) { return hid == HID_SIGMUL ? 4 : 0; }

// One layer.  Software pipeline: the activation of a finished 16-row block is issued, one piece per K step, inside the
// MFMA stream of the next block (the two blocks of a chunk accumulate in two register sets, so nothing is copied);
// A tiles run A_PF tiles ahead in a rotating register queue across block and chunk boundaries; the bias of the next
// block is fetched one block ahead; weight chunks are loaded two chunks ahead into a 3-slot LDS ring (one barrier per
// chunk).  Everything inside a block is straight-line code (a wave-uniform branch per K step costs issue slots and
// fences the scheduler), and its LDS waits are counted (s_waitcnt lgkmcnt(3)), which needs the weight DMA to be issued
// from inline asm: see lds_dma_16.
template <int NB, bool FWD, int KS_IN, int HID, int WAVES, bool HIDDEN, typename Cap, typename NB_T>
__device__ __forceinline__ void run_layer(const NetDesc& net, const LayerDesc L, int l, const char* __restrict__ wpack,
                                          const float* bias_lds, char* wring, opx8 (&Bcur)[KS_REG][NB], NB_T& Bn,
                                          opx8 (&aq)[AQ_LEN], u32x4 (&sgb)[SIG_BUFS][NB], const op_t* stage_wave,
                                          f32x4 (&out)[NB], int wave, int lane, const SigIO& sig, Cap& cap, int& ci) {
    constexpr int PF = A_PF, QN = A_QN;
    constexpr bool BIAS = HID != HID_SIGMUL;   // the reverse sweep has no bias
    const int g = lane >> 4;
    const float* bl = BIAS ? bias_lds + l * BIAS_STRIDE + g * 4 : nullptr;
    const int sig_layer = HID == HID_SIGMUL ? (L.aux & 0xff) - 1 : l;
    const int cap_id = (L.aux >> 8) & 0xff;
    f32x4 accs[2][NB];   // block (c, mbl) accumulates in accs[mbl] while the activation of accs[mbl ^ 1] is issued
    f32x4 bv_next = (f32x4){0, 0, 0, 0};
    if constexpr (BIAS) bv_next = *(const f32x4*)bl;
    auto load_sig = [&](int cc) {   // stored sigmoids of chunk cc (compile-time after unrolling) -> their buffer
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            sgb[sig_slot<HID>(cc)][nb] = *(const u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (cc * NB + nb) * 1024 +
                                                         lane * 16);
    };
#pragma unroll
    for (int c = 0; c < MAX_CHUNKS; ++c) {
        if (c < L.n_chunk) {
            MP_STAMP(0);
#ifndef MP_EXP_DMA_FIRST
            if constexpr (HID == HID_SIGMUL && HIDDEN) load_sig(c);   // HBM stream first, then the (L2-resident) weight stream
#endif
#if !defined(MP_EXP_NOLOAD) && !defined(MP_EXP_SPREAD_DMA)
            if (ci + 2 < net.total_chunks) issue_chunk<KS_IN, WAVES>(wpack, wring, ci + 2, wave, lane);
#endif
#ifdef MP_EXP_DMA_FIRST
            if constexpr (HID == HID_SIGMUL && HIDDEN) load_sig(c);
#endif
            const char* slot = wring + (ci % RING_SLOTS) * chunk_bytes(KS_IN) + lane * 16;
            // chunk ci+1 landed before the previous barrier: its first A tiles are prefetched from this chunk (after the
            // network's last chunk the read hits a stale ring slot and is never used)
            const char* slot_next = wring + ((ci + 1) % RING_SLOTS) * chunk_bytes(KS_IN) + lane * 16;
#pragma unroll
            for (int mbl = 0; mbl < CHUNK_MB; ++mbl) {
#ifdef MP_EXP_PRIO   // ablation: alternate the issue priority of the two waves of a SIMD block by block
                if (((wave >> 2) ^ mbl) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#ifdef MP_EXP_SPREAD_DMA   // ablation: one piece per wave at the start of each block instead of all after the barrier
                if (ci + 2 < net.total_chunks) {
                    if (mbl == 0) issue_chunk<KS_IN, WAVES, 0, 1>(wpack, wring, ci + 2, wave, lane);
                    else issue_chunk<KS_IN, WAVES, 1, 99>(wpack, wring, ci + 2, wave, lane);
                }
#endif
                f32x4 (&acc)[NB] = accs[mbl];
                const f32x4 (&pend)[NB] = accs[mbl ^ 1];   // finished block whose activation is still pending
                const f32x4 bv = bv_next;
                if constexpr (BIAS) bv_next = *(const f32x4*)(bl + (2 * c + mbl + 1) * 16);   // may run 16 floats past the table
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = (FWD && nb > 0) ? (f32x4){0, 0, 0, 0} : bv;
                if constexpr (FWD) {  // half-block layout: only the value half of block 0 carries the bias
                    if (threadIdx.x & 8) acc[0] = (f32x4){0, 0, 0, 0};
                }
                const char* tile = slot + mbl * mb_bytes(KS_IN);
                // pending block = (c, 0) when mbl == 1, (c-1, 1) when mbl == 0
                const bool has_pend = mbl == 1 || c > 0;
                const int pc = mbl == 1 ? c : c - 1, ph = mbl == 1 ? 0 : 1;
#ifdef MP_EXP_NOACT
#define MP_ACT_STMT(KS)
#else
#define MP_ACT_STMT(KS)                                                                                               \
    if constexpr (KS >= act_shift(HID)) {                                                                             \
        if (has_pend)                                                                                                 \
            act_piece<NB, FWD, HID, HIDDEN, KS - act_shift(HID)>(pend, Bn, pc, ph, sgb[sig_slot<HID>(pc)], sig, sig_layer); \
    }
#endif
#ifdef MP_EXP_NOLDS
#define MP_LDS_STMT (void)src;
#else
#define MP_LDS_STMT aq[nk % QN] = *(const opx8*)src;
#endif
#define MP_QSKIP(KS)                                                                                                  \
    {                                                                                                                 \
        constexpr int nk = KS + PF;                                                                                   \
        const char* src = nk < KS_REG ? tile + nk * TILE_BYTES                                                        \
                          : (mbl + 1 < CHUNK_MB ? tile + mb_bytes(KS_IN) + (nk - KS_REG) * TILE_BYTES                 \
                                                : slot_next + (nk - KS_REG) * TILE_BYTES);                            \
        if (nk >= KS_REG) { MP_LDS_STMT }                                                                             \
    }
#define MP_KSTEP(KS)                                                                                                  \
    {                                                                                                                 \
        const opx8 a = aq[KS % QN];                                                                                 \
        {                                                                                                             \
            constexpr int nk = KS + PF;                                                                               \
            const char* src = nk < KS_REG ? tile + nk * TILE_BYTES                                                    \
                              : (mbl + 1 < CHUNK_MB ? tile + mb_bytes(KS_IN) + (nk - KS_REG) * TILE_BYTES             \
                                                    : slot_next + (nk - KS_REG) * TILE_BYTES);                        \
            MP_LDS_STMT                                                                                               \
        }                                                                                                             \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                             \
            acc[nb] = MP_MFMA_F16(a, Bcur[KS][nb], acc[nb], 0, 0, 0);                     \
        MP_ACT_STMT(KS)                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                            \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) {                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                        \
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                                        \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
                if (L.use_reg) {
                    MP_KSTEP(0) MP_KSTEP(1) MP_KSTEP(2) MP_KSTEP(3) MP_KSTEP(4) MP_KSTEP(5) MP_KSTEP(6) MP_KSTEP(7)
                } else {
                    // layer fed only by the encoded input (layer 0): no register K steps; keep the A-tile queue
                    // in step with the tile stream and finish the pending block's activation
                    MP_QSKIP(0) MP_QSKIP(1) MP_QSKIP(2) MP_QSKIP(3) MP_QSKIP(4) MP_QSKIP(5) MP_QSKIP(6) MP_QSKIP(7)
#ifndef MP_EXP_NOACT
                    if (has_pend) act_from<NB, FWD, HID, HIDDEN, 0>(pend, Bn, pc, ph, sgb[sig_slot<HID>(pc)], sig, sig_layer);
#endif
                }
#undef MP_KSTEP
#undef MP_QSKIP
#undef MP_ACT_STMT
#undef MP_LDS_STMT
                if (L.use_in) {
#pragma unroll
                    for (int ks = 0; ks < KS_IN; ++ks) {
                        const opx8 a = *(const opx8*)(tile + (KS_REG + ks) * TILE_BYTES);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const opx8 bi = *(const opx8*)(stage_wave + (nb * 16 + (lane & 15)) * in_stride(KS_IN) +
                                                               ks * 32 + g * 8);
                            acc[nb] = MP_MFMA_F16(a, bi, acc[nb], 0, 0, 0);
                        }
                    }
                }
                if (cap_id == 1) {          // rows 0..47: blocks (0,0), (0,1), (1,0)
                    if (c == 0) cap(1, mbl, acc);
                    else if (c == 1 && mbl == 0) cap(1, 2, acc);
                } else if (cap_id == 2) {   // rows 208..255: blocks (6,1), (7,0), (7,1)
                    if (c == 6 && mbl == 1) cap(2, 0, acc);
                    else if (c == 7) cap(2, 1 + mbl, acc);
                }
                if ((c == 0 || c == MAX_CHUNKS - 1) && mbl == 0) {
                    if (c == L.out_chunk) {  // fp32 rows 0..15 of the out chunk (its layer is linear)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) out[nb] = acc[nb];
                    }
                }
            }
            MP_STAMP(1);
#ifndef MP_EXP_NOBARRIER
            dma_wait_all();   // this wave's pieces of chunk ci+2 (issued a whole chunk ago) and of every earlier chunk
            MP_STAMP(2);
            __syncthreads();  // every wave is done with chunk ci; chunk ci+1 is complete in the ring
            MP_STAMP(3);
#endif
            ++ci;
        }
    }
    // layer ends: drain the pipeline (block (n_chunk - 1, 1); after the chunk's barrier, it touches registers only)
    act_drain<NB, FWD, HID, HIDDEN>(accs[1], Bn, L.n_chunk - 1, sgb, sig, sig_layer);
#pragma unroll
    for (int k = 0; k < KS_REG; ++k)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) Bcur[k][nb] = Bn.get(k, nb);
}

// Runs the whole network for this wave's NB column blocks.
//   Bcur : register K operand of layer 0 (zeros when the network input only enters through the staging tile); on
//          return it holds the last layer's (half) output blocks (e.g. the 256 features).
//   stage_wave : this wave's input staging tile in LDS ([16*NB rows][in_stride] halves, rows = columns): the encoded
//          network input, read on demand as the K operand of K steps 8.. of every layer with use_in.
//   out  : fp32 rows 0..15 of the `out_chunk` (must be the last chunk of its layer).
// The caller must have run prologue() (chunks 0 and 1 in ring slots 0 and 1, barrier).
template <int NB, bool FWD, int KS_IN, int HID, int WAVES, typename Cap = NoCapture>
__device__ __forceinline__ void run_net(const NetDesc& net, const char* __restrict__ wpack, const float* bias_lds,
                                        char* wring, opx8 (&Bcur)[KS_REG][NB], const op_t* stage_wave,
                                        f32x4 (&out)[NB], int wave, int lane, SigIO sig = SigIO{nullptr, 0},
                                        Cap cap = Cap()) {
    int ci = 0;
    // sigmoid fragments of the K steps under construction (HID_SOFTPLUS_SAVE) / about to be applied (HID_SIGMUL): sig_slot
    u32x4 sgb[SIG_BUFS][NB];
#pragma unroll
    for (int b = 0; b < SIG_BUFS; ++b)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) sgb[b][nb] = (u32x4){0u, 0u, 0u, 0u};
    NextB<NB, false> Bn;
    Bn.zero();
    opx8 aq[AQ_LEN];
#pragma unroll
    for (int t = 0; t < A_PF; ++t) aq[t] = *(const opx8*)(wring + t * TILE_BYTES + lane * 16);  // chunk 0, block 0
    // every network on this path is a run of hidden layers followed by its linear output layer(s): two loops instead of
    // a per-layer dispatch keep the two instantiations of the layer body out of each other's register allocation
    int l = 0;
#ifdef MP_EXP_NOPP   // ablation: the interleaved stream of run_layer for every layer
    constexpr bool PP = false;
#else
#ifdef MP_EXP_GRAD_OLD   // ablation: the reverse sweep on the interleaved stream
    constexpr bool PP = !FWD && NB == 2 && HID != HID_SIGMUL;
#else
    constexpr bool PP = !FWD && NB == 2;
#endif
#endif
    static_assert(PP || !SIG8 || (HID != HID_SOFTPLUS_SAVE && HID != HID_SIGMUL),
                  "the interleaved (round-1) stream stores half-precision sigmoids: build the ablation with -DMP_EXP_SIG16");
    if constexpr (PP) {
        // The phase-separated stream keeps THREE chunks in flight (prologue<.., true>: chunks 0, 1 and 2) and fetches
        // chunk ci + 3 behind the barrier that frees chunk ci's slot.
        const int pp_last = net.total_chunks - 1;
        const ChunkMasks cm = chunk_masks(net);
        const DmaLanes<KS_IN> dl = dma_lanes<KS_IN>(__builtin_amdgcn_readfirstlane(wave) & (WAVES / 2 - 1), lane);
        int ring_pos = 0;
        // tile order of the phase-separated stream: (K step, row block); chunk 0's first tiles
#pragma unroll
        for (int t = 0; t < PP_PF; ++t)
            aq[t % PP_QN] = *(const opx8*)(wring + (t % 2) * mb_bytes(KS_IN) + (t / 2) * TILE_BYTES + lane * 16);
        for (; l < net.n_layers && net.layer[l].act != ACT_NONE; ++l)
            run_layer_pp<KS_IN, HID, WAVES, true>(net, net.layer[l], l, wpack, bias_lds, wring, Bcur, Bn, aq, sgb, stage_wave, out,
                                                  wave, lane, sig, cap, ci, ring_pos, cm, dl, pp_last);
        for (; l < net.n_layers; ++l)
            run_layer_pp<KS_IN, HID, WAVES, false>(net, net.layer[l], l, wpack, bias_lds, wring, Bcur, Bn, aq, sgb, stage_wave, out,
                                                   wave, lane, sig, cap, ci, ring_pos, cm, dl, pp_last);
    }
    if constexpr (!PP) {
        for (; l < net.n_layers && net.layer[l].act != ACT_NONE; ++l)
            run_layer<NB, FWD, KS_IN, HID, WAVES, true>(net, net.layer[l], l, wpack, bias_lds, wring, Bcur, Bn, aq, sgb, stage_wave,
                                                        out, wave, lane, sig, cap, ci);
    }
    for (; l < net.n_layers; ++l)
        run_layer<NB, FWD, KS_IN, HID, WAVES, false>(net, net.layer[l], l, wpack, bias_lds, wring, Bcur, Bn, aq, sgb, stage_wave,
                                                     out, wave, lane, sig, cap, ci);
}

// Issues the first two weight chunks into ring slots 0 and 1 and synchronises (also publishes the staging rows).
// THREE: the phase-separated stream (plain-mode kernels) starts with all three ring slots filled.
#ifdef MP_EXP_NOPP
constexpr bool PROLOGUE_THREE = false;
#else
constexpr bool PROLOGUE_THREE = true;
#endif
// Round 6: in two halves.  prologue_issue() goes to the HEAD of a tile, before the tile's inputs are fetched and encoded (the ring is
// free there: every wave has passed the barrier behind the previous tile's last M phase), prologue_wait() behind them -- the first
// chunks' DMA latency then runs beside the worklist -> position loads and the Fourier features instead of after them.
template <int KS_IN, int WAVES, bool THREE = PROLOGUE_THREE>
__device__ __forceinline__ void prologue_issue(const NetDesc& net, const char* __restrict__ wpack, char* wring, int wave, int lane) {
    issue_chunk<KS_IN, WAVES>(wpack, wring, 0, wave, lane);
    if (net.total_chunks > 1) issue_chunk<KS_IN, WAVES>(wpack, wring, 1, wave, lane);
    if constexpr (THREE) {
        if (net.total_chunks > 2) issue_chunk<KS_IN, WAVES>(wpack, wring, 2, wave, lane);
    }
}
__device__ __forceinline__ void prologue_wait() {
    dma_wait_all();
    __syncthreads();
}
template <int KS_IN, int WAVES, bool THREE = PROLOGUE_THREE>
__device__ __forceinline__ void prologue(const NetDesc& net, const char* __restrict__ wpack, char* wring, int wave,
                                         int lane) {
    prologue_issue<KS_IN, WAVES, THREE>(net, wpack, wring, wave, lane);
    prologue_wait();
}
__device__ __forceinline__ void load_bias(const NetDesc& net, const float* __restrict__ bias, float* bias_lds) {
    for (int i = threadIdx.x; i < net.n_layers * BIAS_STRIDE; i += blockDim.x) bias_lds[i] = bias[i];
}

}  // namespace mp
