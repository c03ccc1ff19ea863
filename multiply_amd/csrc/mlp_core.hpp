// Register-resident fused MLP core for gfx950 (CDNA4), bf16 MFMA 16x16x32, fp32 accumulate.
//
// Replaces the per-layer torch.nn.Linear + Softplus/ReLU chain of the reference's
// ImplicitNet.forward (code/lib/model/networks.py:160-181) and RenderingNet.forward
// (networks.py:305-311): all layers of one network are evaluated for a tile of points
// without the activations ever leaving the register file.
//
// Orientation: D[out_feature][point] = W[out][k] * X^T[k][point].
//   A operand = 16x32 weight tile (rows = output features), streamed global -> LDS
//               (global_load_lds, 16 B / lane) in MFMA fragment order, shared by the
//               4 waves of the workgroup.
//   B operand = activations, 32 k-slots x 16 points per (ks, nb); lives in VGPRs.
//   D         = col = lane&15 (point), row = 4*(lane>>4)+reg (output feature).
// Because lane (j,g) receives output rows 4g..4g+3 of every 16-row block and needs
// k-slots 8g..8g+7 of every 32-slot K step of the next layer, the host packs the
// weights with the K permutation
//     slot(ks, g, e) <-> feature 32*ks + (e<4 ? 4g+e : 16+4g+(e-4))
// (see pack_weights.py / mp_pack_weights), so that bf16(act(D)) of output blocks
// (2ks, 2ks+1) IS the B fragment of K step ks: no LDS round trip, no cross-lane moves.
//
// A wave owns NB column blocks of 16 columns.  Plain mode: 16*NB different points.
// Forward-mode (FWD, NB=4): block 0 = values of 16 points, blocks 1..3 = d/dx, d/dy,
// d/dz tangents of the same 16 points, so sdf and its spatial gradient (the normals of
// multiply.py:620-661) come out of one pass: t' = softplus'(z) * (W t).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KS_REG = 8;   // K steps (32 slots each) fed from registers = previous layer output (<=256 feats)
// KS_IN (template parameter, 2 or 3): K steps fed from the encoded network input (<=64 / 96 feats: PE, normals ...)
constexpr int TILE_BYTES = 1024;                  // one 16x32 bf16 A tile, fragment order: [lane][8]
constexpr int CHUNK_MB = 2;                       // 32 output rows = one K step of the next layer
constexpr int MAX_CHUNKS = 9;                     // 8 chunks = 256 rows, +1 "extra output" chunk
constexpr int MAX_LAYERS = 10;
constexpr int BIAS_STRIDE = MAX_CHUNKS * 32;      // 288 floats per layer
__host__ __device__ constexpr int mb_bytes(int ks_in) { return (KS_REG + ks_in) * TILE_BYTES; }  // all K steps of 16 rows
__host__ __device__ constexpr int chunk_bytes(int ks_in) { return CHUNK_MB * mb_bytes(ks_in); }  // 20 / 22 KiB
__host__ __device__ constexpr int in_stride(int ks_in) { return ks_in * 32 + 8; }  // bf16 per staging row (+pad)

enum Act : int { ACT_NONE = 0, ACT_SOFTPLUS = 1, ACT_RELU = 2 };

struct LayerDesc {
    int n_chunk;    // chunks of 32 output rows (1..9)
    int use_reg;    // consume the 8 register K steps
    int use_in;     // consume the 2 input K steps
    int act;        // Act
    int out_chunk;  // chunk whose first 16 rows are returned in fp32 `out` instead of feeding the next layer (-1: none)
};

struct NetDesc {
    int n_layers;
    int total_chunks;
    LayerDesc layer[MAX_LAYERS];
};

// softplus(beta=100, threshold=20) exactly as torch.nn.Softplus (networks.py:85)
__device__ __forceinline__ float softplus100(float z) {
    float t = 100.0f * z;
    // log1p(exp(t))/100 = max(z,0) + log1p(exp(-|t|))/100
    float u = __expf(-fabsf(t));
    float sp = fmaxf(z, 0.0f) + 0.01f * __logf(1.0f + u);
    return t > 20.0f ? z : sp;
}
// d softplus / dz = sigmoid(100 z)
__device__ __forceinline__ float softplus100_grad(float z) {
    float t = 100.0f * z;
    float u = __expf(-fabsf(t));
    float r = __frcp_rn(1.0f + u);
    float s = t >= 0.0f ? r : u * r;
    return t > 20.0f ? 1.0f : s;
}

template <int KS_IN>
__device__ __forceinline__ void issue_chunk(const char* __restrict__ wpack, char* wring, int ci, int wave, int lane) {
    constexpr int CB = chunk_bytes(KS_IN);
    constexpr int NP = CB / TILE_BYTES;   // 1 KiB pieces, dealt round-robin to the 4 waves
    const char* src = wpack + (size_t)ci * CB;
    char* dst = wring + (ci & 1) * CB;
#pragma unroll
    for (int i = 0; i < (NP + 3) / 4; ++i) {
        const int piece = wave + 4 * i;
        if (piece < NP)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + piece * TILE_BYTES + lane * 16),
            (__attribute__((address_space(3))) void*)(dst + piece * TILE_BYTES), 16, 0, 0);
    }
}

template <int NB>
__device__ __forceinline__ void store_bhalf(bf16x8& b, int half, const f32x4& v) {
    if (half == 0) {
        b[0] = (__bf16)v[0]; b[1] = (__bf16)v[1]; b[2] = (__bf16)v[2]; b[3] = (__bf16)v[3];
    } else {
        b[4] = (__bf16)v[0]; b[5] = (__bf16)v[1]; b[6] = (__bf16)v[2]; b[7] = (__bf16)v[3];
    }
}

// Runs the whole network for this wave's NB column blocks.
//   Bcur : register K operand of the first layer that has use_reg (undefined content is fine
//          if layer 0 has use_reg = 0); on return holds the last layer's (bf16) output blocks.
//   Bin  : encoded-input K operand (K steps 8,9 of every layer with use_in).
//   out  : fp32 rows 0..15 of the `out_chunk` of the layer that declares one.
// The caller must have issued chunk 0 into ring slot 0 and synchronised (see prologue()).
template <int NB, bool FWD, int KS_IN>
__device__ __forceinline__ void run_net(const NetDesc& net, const char* __restrict__ wpack, const float* bias_lds,
                                        char* wring, bf16x8 (&Bcur)[KS_REG][NB], const bf16x8 (&Bin)[KS_IN][NB],
                                        f32x4 (&out)[NB], int wave, int lane) {
    const int g = lane >> 4;
    int ci = 0;
    bf16x8 Bnext[KS_REG][NB];
#pragma unroll
    for (int k = 0; k < KS_REG; ++k)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) Bnext[k][nb] = (bf16x8)(__bf16)0.0f;

    for (int l = 0; l < net.n_layers; ++l) {
        const LayerDesc L = net.layer[l];
        const float* bl = bias_lds + l * BIAS_STRIDE;
#pragma unroll
        for (int c = 0; c < MAX_CHUNKS; ++c) {
            if (c < L.n_chunk) {
                if (ci + 1 < net.total_chunks) issue_chunk<KS_IN>(wpack, wring, ci + 1, wave, lane);
                const char* slot = wring + (ci & 1) * chunk_bytes(KS_IN);
#pragma unroll
                for (int mbl = 0; mbl < CHUNK_MB; ++mbl) {
                    f32x4 acc[NB];
                    const f32x4 bv = *(const f32x4*)(bl + c * 32 + mbl * 16 + g * 4);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = (FWD && nb > 0) ? (f32x4){0, 0, 0, 0} : bv;
                    const char* tile = slot + mbl * mb_bytes(KS_IN) + lane * 16;
                    if (L.use_reg) {
#pragma unroll
                        for (int ks = 0; ks < KS_REG; ++ks) {
                            const bf16x8 a = *(const bf16x8*)(tile + ks * TILE_BYTES);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Bcur[ks][nb], acc[nb], 0, 0, 0);
                        }
                    }
                    if (L.use_in) {
#pragma unroll
                        for (int ks = 0; ks < KS_IN; ++ks) {
                            const bf16x8 a = *(const bf16x8*)(tile + (KS_REG + ks) * TILE_BYTES);
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Bin[ks][nb], acc[nb], 0, 0, 0);
                        }
                    }
                    // activation (fp32), then either fp32 out or bf16 K operand of the next layer
                    if (L.act == ACT_SOFTPLUS) {
                        if (FWD) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float z = acc[0][r];
                                const float s = softplus100_grad(z);
                                acc[0][r] = softplus100(z);
#pragma unroll
                                for (int nb = 1; nb < NB; ++nb) acc[nb][r] *= s;
                            }
                        } else {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc[nb][r] = softplus100(acc[nb][r]);
                        }
                    } else if (L.act == ACT_RELU) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[nb][r] = fmaxf(acc[nb][r], 0.0f);
                    }
                    if (c == L.out_chunk) {
                        if (mbl == 0) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) out[nb] = acc[nb];
                        }
                    } else if (c < KS_REG) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) store_bhalf<NB>(Bnext[c < KS_REG ? c : 0][nb], mbl, acc[nb]);
                    }
                }
                __syncthreads();  // every wave is done with chunk ci; chunk ci+1 has landed (vmcnt drained)
                ++ci;
            }
        }
#pragma unroll
        for (int k = 0; k < KS_REG; ++k)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) Bcur[k][nb] = Bnext[k][nb];
    }
}

// Loads the per-layer biases into LDS and the first weight chunk into ring slot 0.
template <int KS_IN>
__device__ __forceinline__ void prologue(const NetDesc& net, const char* __restrict__ wpack, char* wring, int wave,
                                         int lane) {
    issue_chunk<KS_IN>(wpack, wring, 0, wave, lane);
    __syncthreads();
}
__device__ __forceinline__ void load_bias(const NetDesc& net, const float* __restrict__ bias, float* bias_lds) {
    for (int i = threadIdx.x; i < net.n_layers * BIAS_STRIDE; i += blockDim.x) bias_lds[i] = bias[i];
}

}  // namespace mp
