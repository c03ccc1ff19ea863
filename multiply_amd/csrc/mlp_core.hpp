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

template <int KS_IN, int WAVES>
__device__ __forceinline__ void issue_chunk(const char* __restrict__ wpack, char* wring, int ci, int wave, int lane) {
    constexpr int CB = chunk_bytes(KS_IN);
    constexpr int NP = CB / TILE_BYTES;   // 1 KiB pieces, dealt round-robin to the workgroup's waves
    const char* src = wpack + (size_t)ci * CB;
    char* dst = wring + (ci % RING_SLOTS) * CB;
#pragma unroll
    for (int i = 0; i < (NP + WAVES - 1) / WAVES; ++i) {
        const int piece = wave + WAVES * i;
        if (piece < NP)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + piece * TILE_BYTES + lane * 16),
            (__attribute__((address_space(3))) void*)(dst + piece * TILE_BYTES), 16, 0, 0);
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h2 to_h2(float a, float b) { return (h2){(op_t)a, (op_t)b}; }   // one v_cvt_pk_f16_f32 (RTNE)
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
enum Hidden : int { HID_SOFTPLUS = 0, HID_RELU = 1, HID_SOFTPLUS_SAVE = 2, HID_SIGMUL = 3 };
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
__device__ __forceinline__ h2 relu_h2(h2 z) {   // ONE v_pk_max_f16 (the builtin max adds a canonicalising v_pk_max in front)
    unsigned r;
    const unsigned zi = bits(z);
    asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(zi));
    return __builtin_bit_cast(h2, r);
}
__device__ __forceinline__ h2 softplus2(h2 z) {   // h' = max(z',0) + log2(1 + 2^-|z'|)
#ifdef MP_EXP_NOTRANS
    return relu_h2(z) + z * (h2){(op_t)0.001f, (op_t)0.001f};
#else
    const h2 u = exp2_neg_abs_h2(z);
    return relu_h2(z) + log2_h2(u + (h2){(op_t)1.0f, (op_t)1.0f});
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
// of the next block's MFMA stream.  `hidden` (wave-uniform): apply the nonlinearity, else pass through.
//   plain  : q = 0..3 -> column block q/2, row pair q%2
//   forward: half-block tangent layout (8 points per wave): block 0 = [values | d/dx], block 1 = [d/dy | d/dz]; lanes
//            with (lane & 8) == 0 hold the value / d/dy columns of point lane&7, the others d/dx / d/dz.
//            q = 0, 1: row pair q of both blocks (softplus + sigmoid on the values, tangents scaled by the sigmoid)
template <int NB, bool FWD, int HID, int q, typename NB_T>
__device__ __forceinline__ void act_piece(f32x4 (&p)[NB], bool hidden, NB_T& Bn, int pc, int ph, u32x4 (&sg)[NB],
                                          const SigIO& sig, int sig_layer) {
    static_assert(NB == 2, "the MLP core is specialised for 2 column blocks per wave (2 waves per SIMD)");
    if constexpr (FWD) {
        if constexpr (q < 2) {
            h2 z = to_h2(p[0][2 * q], p[0][2 * q + 1]);
            h2 t = to_h2(p[1][2 * q], p[1][2 * q + 1]);
            if (hidden) {
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
            if constexpr (HID == HID_SOFTPLUS_SAVE) {
                if (hidden) {
                    const h2 h = softplus2(z);
                    const unsigned sv = bits(exp2_h2(z - h));   // sigmoid(z') = 2^(z' - h')
                    if (ph == 0) { if (j == 0) sg[nb][0] = sv; else sg[nb][1] = sv; }
                    else { if (j == 0) sg[nb][2] = sv; else sg[nb][3] = sv; }
                    z = h;
                    if (ph == 1 && j == 1 && pc < KS_REG)
                        *(u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (pc * NB + nb) * 1024 + (threadIdx.x & 63) * 16) = sg[nb];
                }
            } else if constexpr (HID == HID_SIGMUL) {
                if (hidden) {
                    const unsigned sv = ph == 0 ? (j == 0 ? sg[nb][0] : sg[nb][1]) : (j == 0 ? sg[nb][2] : sg[nb][3]);
                    z = z * __builtin_bit_cast(h2, sv);
                }
            } else {
                if (hidden) z = HID == HID_SOFTPLUS ? softplus2(z) : relu_h2(z);
            }
            if (pc < KS_REG) Bn.put(pc, nb, ph, j, z);
        }
    }
}

template <int NB, bool FWD, int HID, int q, typename NB_T>
__device__ __forceinline__ void act_from(f32x4 (&p)[NB], bool hidden, NB_T& Bn, int pc, int ph, u32x4 (&sg)[NB],
                                         const SigIO& sig, int sig_layer) {
    act_piece<NB, FWD, HID, q>(p, hidden, Bn, pc, ph, sg, sig, sig_layer);
    if constexpr (q + 1 < 8) act_from<NB, FWD, HID, q + 1>(p, hidden, Bn, pc, ph, sg, sig, sig_layer);
}

// Runs the whole network for this wave's NB column blocks.
//   Bcur : register K operand of layer 0 (zeros when the network input only enters through the staging tile); on
//          return it holds the last layer's (half) output blocks (e.g. the 256 features).
//   stage_wave : this wave's input staging tile in LDS ([16*NB rows][in_stride] halves, rows = columns): the encoded
//          network input, read on demand as the K operand of K steps 8.. of every layer with use_in.
//   out  : fp32 rows 0..15 of the `out_chunk` (must be the last chunk of its layer).
// Software pipeline: the activation of a finished 16-row block is issued, one piece per K step, inside the MFMA stream
// of the next block; A tiles run PF tiles ahead in a rotating register queue across block and chunk boundaries;
// weight chunks are loaded two chunks ahead into a 3-slot LDS ring (one barrier per chunk).
// The caller must have run prologue() (chunks 0 and 1 in ring slots 0 and 1, barrier).
struct NoCapture {
    template <int NB>
    __device__ __forceinline__ void operator()(int, int, const f32x4 (&)[NB]) const {}
};

template <int NB, bool FWD, int KS_IN, int HID, int WAVES, typename Cap = NoCapture>
__device__ __forceinline__ void run_net(const NetDesc& net, const char* __restrict__ wpack, const float* bias_lds,
                                        char* wring, opx8 (&Bcur)[KS_REG][NB], const op_t* stage_wave,
                                        f32x4 (&out)[NB], int wave, int lane, SigIO sig = SigIO{nullptr, 0},
                                        Cap cap = Cap()) {
    const int g = lane >> 4;
    int ci = 0;
    // sigmoid fragments of the K steps under construction (HID_SOFTPLUS_SAVE / HID_SIGMUL), double-buffered by chunk
    // parity: chunk c's group is live from block (c,1) to block (c+1,0), the next one is loaded at chunk c+1's start
    u32x4 sgb[2][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) sgb[0][nb] = sgb[1][nb] = (u32x4){0u, 0u, 0u, 0u};
    NextB<NB, false> Bn;
    Bn.zero();
    constexpr int PF = 3, QN = 4;   // prefetch distance / queue length in A tiles
    opx8 aq[QN];
#pragma unroll
    for (int t = 0; t < PF; ++t) aq[t] = *(const opx8*)(wring + t * TILE_BYTES + lane * 16);  // chunk 0, block 0
    for (int l = 0; l < net.n_layers; ++l) {
        const LayerDesc L = net.layer[l];
        const float* bl = bias_lds + l * BIAS_STRIDE;
        const bool hidden = L.act != ACT_NONE;
        const int sig_layer = HID == HID_SIGMUL ? (L.aux & 0xff) - 1 : l;
        const int cap_id = (L.aux >> 8) & 0xff;
        f32x4 pend[NB];  // finished block whose activation is still pending
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) pend[nb] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            if (c < L.n_chunk) {
#ifndef MP_EXP_NOLOAD
                if (ci + 2 < net.total_chunks) issue_chunk<KS_IN, WAVES>(wpack, wring, ci + 2, wave, lane);
#endif
                const char* slot = wring + (ci % RING_SLOTS) * chunk_bytes(KS_IN) + lane * 16;
                // chunk ci+1 landed before the previous barrier: its first A tiles are prefetched from this chunk
                const char* slot_next = wring + ((ci + 1) % RING_SLOTS) * chunk_bytes(KS_IN) + lane * 16;
                const bool has_next = ci + 1 < net.total_chunks;
                if constexpr (HID == HID_SIGMUL) {
                    if (hidden) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            sgb[c & 1][nb] = *(const u32x4*)(sig.base + (size_t)sig_layer * sig.layer_bytes + (c * NB + nb) * 1024 +
                                                             lane * 16);
                    }
                }
#pragma unroll
                for (int mbl = 0; mbl < CHUNK_MB; ++mbl) {
                    f32x4 acc[NB];
                    const f32x4 bv = *(const f32x4*)(bl + c * 32 + mbl * 16 + g * 4);
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
#define MP_ACT_STMT(KS) if (has_pend) act_piece<NB, FWD, HID, KS>(pend, hidden, Bn, pc, ph, sgb[pc & 1], sig, sig_layer);
#endif
#ifdef MP_EXP_NOLDS
#define MP_LDS_STMT (void)src;
#else
#define MP_LDS_STMT if (nk < KS_REG || mbl + 1 < CHUNK_MB || has_next) aq[nk % QN] = *(const opx8*)src;
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
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, Bcur[KS][nb], acc[nb], 0, 0, 0);                     \
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
                        if (has_pend) act_from<NB, FWD, HID, 0>(pend, hidden, Bn, pc, ph, sgb[pc & 1], sig, sig_layer);
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
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bi, acc[nb], 0, 0, 0);
                            }
                        }
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) pend[nb] = acc[nb];
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
                if (c == L.n_chunk - 1) {  // layer ends: drain the pipeline (block (c, 1))
                    act_from<NB, FWD, HID, 0>(pend, hidden, Bn, c, 1, sgb[c & 1], sig, sig_layer);
                }
#ifndef MP_EXP_NOBARRIER
                __syncthreads();  // every wave is done with chunk ci; chunk ci+2's loads have had a whole chunk to land
#endif
                ++ci;
            }
        }
#pragma unroll
        for (int k = 0; k < KS_REG; ++k)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) Bcur[k][nb] = Bn.get(k, nb);
    }
}

// Issues the first two weight chunks into ring slots 0 and 1 and synchronises (also publishes the staging rows).
template <int KS_IN, int WAVES>
__device__ __forceinline__ void prologue(const NetDesc& net, const char* __restrict__ wpack, char* wring, int wave,
                                         int lane) {
    issue_chunk<KS_IN, WAVES>(wpack, wring, 0, wave, lane);
    if (net.total_chunks > 1) issue_chunk<KS_IN, WAVES>(wpack, wring, 1, wave, lane);
    __syncthreads();
}
__device__ __forceinline__ void load_bias(const NetDesc& net, const float* __restrict__ bias, float* bias_lds) {
    for (int i = threadIdx.x; i < net.n_layers * BIAS_STRIDE; i += blockDim.x) bias_lds[i] = bias[i];
}

}  // namespace mp
