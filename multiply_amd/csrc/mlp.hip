// Fused MLP kernels built on mlp_core.hpp.  Entry points: see include/multiply_hip.h.
//   mp_mlp_sdf    foreground ImplicitNet, sdf column only (sampler queries; ray_sampler.py:85-88)
//   mp_mlp_full   ImplicitNet, all outputs (query_oc callers; multiply_model.py:941-945)
//   mp_mlp_shade  foreground ImplicitNet in forward mode: sdf, d sdf/d x_c, features (multiply.py:643-661)
//   mp_mlp_color  foreground RenderingNet 'pose_no_view' (networks.py:277-281)
//   mp_background NeRF++ background branch (multiply.py:514-539, 682-726)
// Geometry: plain kernels run 8 waves x (2 column blocks of 16 points) = 2 waves per SIMD; the forward-mode kernel needs
// value and its three tangents in one wave: 8 waves x 8 points in a half-block layout, also 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include "common.hpp"
#include "../../include/multiply_hip.h"
#include "mlp_core.hpp"

using namespace mp;

namespace {

static_assert(sizeof(MpNet) == sizeof(NetDesc), "host / device net descriptors must match");
static_assert(MP_BIAS_STRIDE == BIAS_STRIDE && MP_MAX_LAYERS == MAX_LAYERS && MP_MAX_CHUNKS == MAX_CHUNKS, "abi");

constexpr int BIAS_BYTES = MAX_LAYERS * BIAS_STRIDE * 4;

template <int KS_IN, int NB, int WAVES>
struct Lds {
    static constexpr int PTS = 16 * NB;          // staging rows (= columns) per wave
    static constexpr int TILE = PTS * WAVES;     // columns per workgroup pass
    static constexpr int ring = 0;
    static constexpr int bias0 = RING_SLOTS * chunk_bytes(KS_IN);
    static constexpr int bias1 = bias0 + BIAS_BYTES;
    static constexpr int stage = bias1 + BIAS_BYTES;
    static constexpr int scratch = stage + WAVES * PTS * in_stride(KS_IN) * 2;
    static constexpr int total = scratch + WAVES * PTS * 16;  // 4 floats per column
};

// Fourier features of a D-vector into one staging row (half): [x, sin(2^0 x), cos(2^0 x), ...]  (embedders.py)
template <int D, int L, int KS_IN>
__device__ __forceinline__ void stage_pe(op_t* row, const float (&x)[D]) {
    constexpr int NF = D + 2 * D * L;
    static_assert(NF <= KS_IN * 32, "encoding does not fit the input K steps");
#pragma unroll
    for (int a = 0; a < D; ++a) row[a] = (op_t)x[a];
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float s, c;
        sincosf(x[a], &s, &c);
#pragma unroll
        for (int k = 0; k < L; ++k) {
            row[D + 2 * D * k + a] = (op_t)s;
            row[D + 2 * D * k + D + a] = (op_t)c;
            const float s2 = 2.0f * s * c, c2 = 1.0f - 2.0f * s * s;  // angle doubling: next octave
            s = s2;
            c = c2;
        }
    }
#pragma unroll
    for (int f = NF; f < KS_IN * 32; ++f) row[f] = (op_t)0.0f;
}

// d/dx_axis of the 3-D, L-octave Fourier features (tangent row for forward mode)
template <int L, int KS_IN>
__device__ __forceinline__ void stage_pe_tangent(op_t* row, const float (&x)[3], int axis) {
    constexpr int D = 3;
#pragma unroll
    for (int f = 0; f < KS_IN * 32; ++f) row[f] = (op_t)0.0f;
    float s, c;
    const float xa = axis == 0 ? x[0] : (axis == 1 ? x[1] : x[2]);
    sincosf(xa, &s, &c);
    row[axis] = (op_t)TANGENT_SCALE;
    float f = TANGENT_SCALE;
#pragma unroll
    for (int k = 0; k < L; ++k) {
        row[D + 2 * D * k + axis] = (op_t)(f * c);
        row[D + 2 * D * k + D + axis] = (op_t)(-f * s);
        const float s2 = 2.0f * s * c, c2 = 1.0f - 2.0f * s * s;
        s = s2;
        c = c2;
        f *= 2.0f;
    }
}

template <int NB>
__device__ __forceinline__ void zero_b(opx8 (&B)[KS_REG][NB]) {
#pragma unroll
    for (int k = 0; k < KS_REG; ++k)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) B[k][nb] = (opx8)(op_t)0.0f;
}

// ------------------------------------------------------------------------------------------------ sdf only
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mlp_sdf(const NetDesc net, const char* __restrict__ wpack,
                                                        const float* __restrict__ bias, const float* __restrict__ xc,
                                                        const int* __restrict__ worklist,
                                                        const int* __restrict__ count_p, int max_count,
                                                        float* __restrict__ sdf_out) {
    constexpr int KS_IN = 2;
    using L = Lds<KS_IN, NB, WAVES>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int count = count_p ? min(*count_p, max_count) : max_count;
    float* bias_lds = (float*)(smem + L::bias0);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * in_stride(KS_IN);
    load_bias(net, bias, bias_lds);
    for (int t = blockIdx.x; t * L::TILE < count; t += gridDim.x) {
        MP_STAMP_AT(HID_SOFTPLUS, 120, 0);
        prologue_issue<KS_IN, WAVES>(net, wpack, smem + L::ring, wave, lane);
        const int w = t * L::TILE + wave * L::PTS + lane;
        const int id = (lane < L::PTS && w < count) ? (worklist ? worklist[w] : w) : -1;
        if (lane < L::PTS) {
            float x[3] = {0.f, 0.f, 0.f};
            if (id >= 0) { x[0] = xc[3 * (size_t)id]; x[1] = xc[3 * (size_t)id + 1]; x[2] = xc[3 * (size_t)id + 2]; }
            stage_pe<3, 6, KS_IN>(stage + lane * in_stride(KS_IN), x);
        }
        MP_STAMP_AT(HID_SOFTPLUS, 120, 1);
        opx8 Bcur[KS_REG][NB];
        f32x4 out[NB];
        zero_b<NB>(Bcur);
        prologue_wait();  // barrier inside: staging rows visible
        MP_STAMP_AT(HID_SOFTPLUS, 120, 2);
        run_net<NB, false, KS_IN, HID_SOFTPLUS, WAVES>(net, wpack, bias_lds, smem + L::ring, Bcur, stage, out, wave, lane);
        MP_STAMP_AT(HID_SOFTPLUS, 120, 3);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int pid = __shfl(id, nb * 16 + (lane & 15));
            if (lane < 16 && pid >= 0) sdf_out[pid] = out[nb][0];
        }
        MP_STAMP_AT(HID_SOFTPLUS, 121, 0);
    }
}

// ------------------------------------------------------------------------------------------------ sdf only, split activations
// The sampler's queries at an order of magnitude below the half-precision kernel's error, for twice its time (round 6; DESIGN.md
// section 4): the SAME packed half-precision weights and bias table as k_mlp_sdf, but a wave carries 16 points whose activations
// travel as two halves, x = hi + lo, in its two column blocks (mlp_core.hpp HID_SOFTPLUS_X2): W_h x_h + W_h x_l, fp32 softplus.
// Fourier features from sincosf per octave in fp32 (the half-precision kernel doubles the angle), split like the activations.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mlp_sdf_x2(const NetDesc net, const char* __restrict__ wpack,
                                                           const float* __restrict__ bias, const float* __restrict__ xc,
                                                           const int* __restrict__ worklist,
                                                           const int* __restrict__ count_p, int max_count,
                                                           float* __restrict__ sdf_out) {
    constexpr int KS_IN = 2, NB = 2, WPTS = 16, TILE = WPTS * WAVES;
    using L = Lds<KS_IN, NB, WAVES>;               // 32 staging rows per wave: rows 0..15 = hi, 16..31 = lo of its 16 points
    constexpr int NF = 39, STR = in_stride(KS_IN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const int count = count_p ? min(*count_p, max_count) : max_count;
    float* bias_lds = (float*)(smem + L::bias0);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * STR;
    load_bias(net, bias, bias_lds);
    for (int t = blockIdx.x; t * TILE < count; t += gridDim.x) {
        prologue_issue<KS_IN, WAVES>(net, wpack, smem + L::ring, wave, lane);
        const int w = t * TILE + wave * WPTS + j;
        const int id = w < count ? (worklist ? worklist[w] : w) : -1;    // every lane knows the id of point lane & 15
        {
            op_t* rh = stage + j * STR;
            op_t* rl = stage + (16 + j) * STR;
            auto put = [&](int f, float v) {
                const op_t hi = (op_t)v;
                rh[f] = hi;
                rl[f] = (op_t)(v - (float)hi);
            };
            if (g < 3) {             // lanes of group g: axis g of point j -- x, then sin / cos of x 2^k (embedders.py layout)
                const float x = id >= 0 ? xc[3 * (size_t)id + g] : 0.0f;
                put(g, x);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    float sn, cs;
                    sincosf(x * (float)(1 << k), &sn, &cs);
                    put(3 + 6 * k + g, sn);
                    put(3 + 6 * k + 3 + g, cs);
                }
            } else {
#pragma unroll
                for (int f = NF; f < KS_IN * 32; ++f) { rh[f] = (op_t)0.0f; rl[f] = (op_t)0.0f; }
            }
        }
        opx8 Bcur[KS_REG][NB];
        f32x4 out[NB];
        zero_b<NB>(Bcur);
        prologue_wait();  // barrier inside: staging rows visible
        run_net<NB, false, KS_IN, HID_SOFTPLUS_X2, WAVES>(net, wpack, bias_lds, smem + L::ring, Bcur, stage, out, wave, lane);
        if (lane < 16 && id >= 0) sdf_out[id] = out[0][0] + out[1][0];
    }
}

// ------------------------------------------------------------------------------------------------ all outputs
template <int D_IN, int LFREQ, int KS_IN, int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mlp_full(const NetDesc net, const char* __restrict__ wpack,
                                                         const float* __restrict__ bias, const float* __restrict__ x,
                                                         int n, float* __restrict__ outp) {
    using L = Lds<KS_IN, NB, WAVES>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* bias_lds = (float*)(smem + L::bias0);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * in_stride(KS_IN);
    load_bias(net, bias, bias_lds);
    for (int t = blockIdx.x; t * L::TILE < n; t += gridDim.x) {
        const int id0 = t * L::TILE + wave * L::PTS;
        if (lane < L::PTS) {
            const int id = id0 + lane < n ? id0 + lane : -1;
            float xi[D_IN];
#pragma unroll
            for (int a = 0; a < D_IN; ++a) xi[a] = id >= 0 ? x[(size_t)id * D_IN + a] : 0.f;
            stage_pe<D_IN, LFREQ, KS_IN>(stage + lane * in_stride(KS_IN), xi);
        }
        opx8 Bcur[KS_REG][NB];
        f32x4 out[NB];
        zero_b<NB>(Bcur);
        prologue<KS_IN, WAVES>(net, wpack, smem + L::ring, wave, lane);
        run_net<NB, false, KS_IN, HID_SOFTPLUS, WAVES>(net, wpack, bias_lds, smem + L::ring, Bcur, stage, out, wave, lane);
        const int j = lane & 15, g = lane >> 4;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int pid = id0 + nb * 16 + j;
            if (pid < n) {
                float* o = outp + (size_t)pid * 257;
                if (g == 0) o[0] = out[nb][0];
#pragma unroll
                for (int ks = 0; ks < KS_REG; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e)  // inverse of the K-slot permutation (mlp_core.hpp header)
                        o[1 + 32 * ks + (e < 4 ? 4 * g + e : 16 + 4 * g + e - 4)] = (float)Bcur[ks][nb][e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ shading: fwd mode
// 8 waves (2 per SIMD), each 8 points in the half-block tangent layout of mlp_core.hpp:
//   block 0 = [values of points 0..7 | d/dx], block 1 = [d/dy | d/dz].
// Feature fragments are written for tiles of 64 work items in the colour kernel's 16-column block layout:
//   feat_frag[tile][ks][block = wave/2][lane' = (col + 8*(wave&1)) + 16 g][8 halves]
__global__ __launch_bounds__(512) void k_mlp_shade(const NetDesc net, const char* __restrict__ wpack,
                                                   const float* __restrict__ bias, const float* __restrict__ xc,
                                                   const float* __restrict__ jinv, const int* __restrict__ worklist,
                                                   const int* __restrict__ count_p, int max_count,
                                                   float* __restrict__ sdf_out, float* __restrict__ normal_out,
                                                   char* __restrict__ feat_frag) {
    constexpr int KS_IN = 2, NB = 2, WAVES = 8;
    using L = Lds<KS_IN, NB, WAVES>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int count = count_p ? min(*count_p, max_count) : max_count;
    float* bias_lds = (float*)(smem + L::bias0);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * in_stride(KS_IN);
    load_bias(net, bias, bias_lds);
    for (int t = blockIdx.x; t * 64 < count; t += gridDim.x) {
        // staging: lane l < 32 builds column l: block l>>4, column l&15 -> point (l&7), role 2*(l>>4) + ((l>>3)&1)
        const int pt = lane & 7, role = 2 * ((lane >> 4) & 1) + ((lane >> 3) & 1);
        const int w = t * 64 + wave * 8 + pt;
        const int id = w < count ? (worklist ? worklist[w] : w) : -1;   // every lane knows the id of point lane&7
        if (lane < 32) {
            float x[3] = {0.f, 0.f, 0.f};
            if (id >= 0) { x[0] = xc[3 * (size_t)id]; x[1] = xc[3 * (size_t)id + 1]; x[2] = xc[3 * (size_t)id + 2]; }
            op_t* row = stage + lane * in_stride(KS_IN);
            if (role == 0) stage_pe<3, 6, KS_IN>(row, x);
            else stage_pe_tangent<6, KS_IN>(row, x, role - 1);
        }
        opx8 Bcur[KS_REG][NB];
        f32x4 out[NB];
        zero_b<NB>(Bcur);
        prologue<KS_IN, WAVES, false>(net, wpack, smem + L::ring, wave, lane);
        run_net<NB, true, KS_IN, HID_SOFTPLUS, WAVES>(net, wpack, bias_lds, smem + L::ring, Bcur, stage, out, wave, lane);
        if ((lane & 8) == 0) {   // value columns: features of point lane&7
            const int lp = ((lane & 15) + 8 * (wave & 1)) + 16 * (lane >> 4);
#pragma unroll
            for (int ks = 0; ks < KS_REG; ++ks)
                *(opx8*)(feat_frag + (((size_t)t * KS_REG + ks) * 4 + (wave >> 1)) * 1024 + lp * 16) = Bcur[ks][0];
        }
        // row 0 of the last layer: lanes 0..7 hold sdf (block 0) and d/dy (block 1), lanes 8..15 d/dx and d/dz
        // (tangent columns are carried at TANGENT_SCALE; the normalisation below is scale-free but for its eps clamps)
        constexpr float TS_INV = 1.0f / TANGENT_SCALE;
        const float gx = TS_INV * __shfl(out[0][0], (lane & 7) + 8), gz = TS_INV * __shfl(out[1][0], (lane & 7) + 8);
        if (lane < 8 && id >= 0) {
            const float gy = TS_INV * out[1][0];
            const float* Ji = jinv + 9 * (size_t)id;
            float n0 = gx * Ji[0] + gy * Ji[3] + gz * Ji[6];
            float n1 = gx * Ji[1] + gy * Ji[4] + gz * Ji[7];
            float n2 = gx * Ji[2] + gy * Ji[5] + gz * Ji[8];
            float inv = 1.0f / fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-12f);  // F.normalize default eps
            n0 *= inv; n1 *= inv; n2 *= inv;
            inv = 1.0f / fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-6f);         // multiply.py:606
            sdf_out[id] = out[0][0];
            normal_out[3 * (size_t)id] = n0 * inv;
            normal_out[3 * (size_t)id + 1] = n1 * inv;
            normal_out[3 * (size_t)id + 2] = n2 * inv;
        }
    }
}

// ------------------------------------------------------------------------------------------------ shading: reverse mode
// sdf, features and d sdf / d x_c in TWO sweeps of one network column each (the forward-mode kernel above pushes four
// columns -- value and three tangents -- through the network):
//   sweep 1  (k_mlp_fwdsave) plain forward pass (32 points per wave); sigmoid(100 z) of every hidden unit is written out,
//            4 KiB per point (f16, operand-fragment layout), for a SEGMENT of the worklist at a time (bounded buffer).
//            [One fused kernel doing both sweeps per tile out of a cache-resident 1 MiB block was tried and is slower
//            (96.7 vs 74.6 ms/frame).]
//   sweep 2  (k_mlp_grad) reverse sweep through the TRANSPOSED layers (hip.py implicit_grad_plans):
//            V_7 = sigma'_7 (.) W_8[sdf row];  V_{l-1} = sigma'_{l-1} (.) (W_l^T V_l);  the "activation" of the shared core is the
//            multiplication by the stored sigmoid.  The rows of W_4^T and W_0^T that belong to the Fourier-feature inputs
//            are contracted on the fly with d PE / d x (two small per-wave LDS tables) and give the gradient;
//            normal = normalize(normalize(grad . Jinv))  (multiply.py:606, 661).
struct GradCapture {
    const op_t* tab_a;   // this wave's [32 points][48]: d PE_f / d x for the rows 0..47 of the last reverse layer (f = row)
    const op_t* tab_b;   // ... for rows 208..255 of the skip layer's transpose (f = row - 217, 0 where row < 217)
    // Encoding feature f differentiates along axis f mod 3 (embedders.py layout: x, then per octave sin(3), cos(3)), and
    // this lane's row of block bi, register r is f = 16 bi + 4 (lane >> 4) + r (- 9 in the skip layer), so its axis is
    // (bi + r + (lane >> 4)) mod 3: accumulate by the COMPILE-TIME part k = (bi + r) mod 3 and undo the per-lane rotation
    // once per tile (unrotate) -- no per-axis selects (their lane masks, hoisted, used to cost ~140 SGPRs).
    float (&g)[2][3];
    template <int NB>
    __device__ __forceinline__ void operator()(int cid, int bi, const f32x4 (&acc)[NB]) const {
        const int lane = threadIdx.x & 63, j = lane & 15, gq = lane >> 4;
        const op_t* tab = cid == 1 ? tab_a : tab_b;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const op_t* tp = tab + (16 * nb + j) * 48 + 16 * bi + 4 * gq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[nb][r] * (float)tp[r];
                const int k = (bi + r) % 3;
                g[nb][0] += k == 0 ? v : 0.0f;
                g[nb][1] += k == 1 ? v : 0.0f;
                g[nb][2] += k == 2 ? v : 0.0f;
            }
        }
    }
    // rotated accumulator k holds axis (k + lane>>4) mod 3
    static __device__ __forceinline__ void unrotate(float (&g)[2][3]) {
        const int s = ((threadIdx.x & 63) >> 4) % 3;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float a0 = g[nb][0], a1 = g[nb][1], a2 = g[nb][2];
            g[nb][0] = s == 0 ? a0 : (s == 1 ? a2 : a1);
            g[nb][1] = s == 0 ? a1 : (s == 1 ? a0 : a2);
            g[nb][2] = s == 0 ? a2 : (s == 1 ? a1 : a0);
        }
    }
};


// ---- sweep 1: work indices [offset, offset + seg) of the worklist.  sig block of (tile t, wave w) of the segment:
//      sig + ((t * 8 + w) * 8 layers) * SIG_LAYER;  inside: [layer][K step = chunk][lane][16 B] as UNORM8 (both column blocks of
//      a chunk in one 16-byte vector; half-precision build: [layer][K step][block][lane][16 B])
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mlp_fwdsave(const NetDesc net, const char* __restrict__ wpack,
                                                            const float* __restrict__ bias, const float* __restrict__ xc,
                                                            const int* __restrict__ worklist,
                                                            const int* __restrict__ count_p, int max_count, int offset,
                                                            int seg, float* __restrict__ sdf_out,
                                                            char* __restrict__ feat_frag, char* __restrict__ sigbuf) {
    constexpr int KS_IN = 2;
    using L = Lds<KS_IN, NB, WAVES>;
    static_assert(L::PTS == 32 && WAVES == 8, "feature / sigmoid block addressing assumes 8 waves x 32 points");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int count = min(count_p ? min(*count_p, max_count) : max_count, offset + seg);
    if (offset >= count) return;
    float* bias_lds = (float*)(smem + L::bias0);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * in_stride(KS_IN);
    load_bias(net, bias, bias_lds);
    constexpr int SIG_LAYER = KS_REG * SIG_CHUNK_BYTES;   // one wave's sigmoids of one layer: 8 (UNORM8) or 16 KiB
    for (int t = blockIdx.x; offset + t * L::TILE < count; t += gridDim.x) {
        prologue_issue<KS_IN, WAVES>(net, wpack, smem + L::ring, wave, lane);
        const int w = offset + t * L::TILE + wave * L::PTS + lane;
        const int id = (lane < L::PTS && w < count) ? (worklist ? worklist[w] : w) : -1;
        if (lane < L::PTS) {
            float x[3] = {0.f, 0.f, 0.f};
            if (id >= 0) { x[0] = xc[3 * (size_t)id]; x[1] = xc[3 * (size_t)id + 1]; x[2] = xc[3 * (size_t)id + 2]; }
            stage_pe<3, 6, KS_IN>(stage + lane * in_stride(KS_IN), x);
        }
        opx8 Bcur[KS_REG][NB];
        f32x4 out[NB];
        zero_b<NB>(Bcur);
        prologue_wait();
        MP_STAMP_AT(HID_SOFTPLUS_SAVE, 120, 2);
        #ifdef MP_EXP_SIGCACHED   // ablation: every tile uses the first workgroup-slots of the buffer (cache resident)
        const SigIO sio = {sigbuf + ((size_t)(blockIdx.x) * WAVES + wave) * (size_t)(8 * SIG_LAYER), SIG_LAYER};
#else
        const SigIO sio = {sigbuf + ((size_t)t * WAVES + wave) * (size_t)(8 * SIG_LAYER), SIG_LAYER};
#endif
        run_net<NB, false, KS_IN, HID_SOFTPLUS_SAVE, WAVES>(net, wpack, bias_lds, smem + L::ring, Bcur, stage, out, wave, lane,
                                                            sio);
        MP_STAMP_AT(HID_SOFTPLUS_SAVE, 120, 3);
        // features in the colour kernel's layout: tiles of 64 work items (offset is a multiple of 256), 4 blocks of 16 columns
        const size_t tile = (size_t)(offset / 64) + (size_t)t * 4 + (wave >> 1);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int ks = 0; ks < KS_REG; ++ks)
                *(opx8*)(feat_frag + ((tile * KS_REG + ks) * 4 + (wave & 1) * 2 + nb) * 1024 + lane * 16) = Bcur[ks][nb];
            const int pid = __shfl(id, nb * 16 + (lane & 15));
            if (lane < 16 && pid >= 0) sdf_out[pid] = out[nb][0];
        }
    }
}

// ---- sweep 2
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mlp_grad(const NetDesc net, const char* __restrict__ wpack,
                                                         const op_t* __restrict__ w8_slots, const float* __restrict__ xc,
                                                         const float* __restrict__ jinv, const int* __restrict__ worklist,
                                                         const int* __restrict__ count_p, int max_count, int offset, int seg,
                                                         const char* __restrict__ sigbuf, float* __restrict__ normal_out) {
    constexpr int KS_IN = 0, PTS = 16 * NB, TILE = PTS * WAVES;   // no input-fed K steps: 16 KiB weight chunks
    static_assert(NB == 2 && WAVES == 8, "matches k_mlp_fwdsave's tile / wave / point addressing");
    constexpr int RING = RING_SLOTS * chunk_bytes(KS_IN);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gq = lane >> 4;
    const int count = min(count_p ? min(*count_p, max_count) : max_count, offset + seg);
    if (offset >= count) return;
    op_t* w8 = (op_t*)(smem + RING);                                  // [256] sdf-row weights in K-slot order
    op_t* tabs = w8 + 256 + wave * (2 * PTS * 48);
    for (int i = threadIdx.x; i < 256; i += blockDim.x) w8[i] = w8_slots[i];
    constexpr int SIG_LAYER = KS_REG * SIG_CHUNK_BYTES;   // one wave's sigmoids of one layer: 8 (UNORM8) or 16 KiB
    for (int t = blockIdx.x; offset + t * TILE < count; t += gridDim.x) {
        const int w = offset + t * TILE + wave * PTS + (lane & (PTS - 1));
        const int id = w < count ? (worklist ? worklist[w] : w) : -1;     // lanes l and l+32 both know point l's id
        if (lane < PTS) {   // d PE_f / d x_axis(f), f = 0..38 (embedders.py layout: x, then per octave sin(3), cos(3))
            float x[3] = {0.f, 0.f, 0.f};
            if (id >= 0) { x[0] = xc[3 * (size_t)id]; x[1] = xc[3 * (size_t)id + 1]; x[2] = xc[3 * (size_t)id + 2]; }
            op_t* ta = tabs + lane * 48;
            op_t* tb = tabs + PTS * 48 + lane * 48;
#pragma unroll
            for (int f = 0; f < 48; ++f) { ta[f] = (op_t)0.0f; tb[f] = (op_t)0.0f; }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ta[a] = (op_t)1.0f;
                float sn, cs, fr = 1.0f;
                sincosf(x[a], &sn, &cs);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    ta[3 + 6 * k + a] = (op_t)(fr * cs);
                    ta[3 + 6 * k + 3 + a] = (op_t)(-fr * sn);
                    const float s2 = 2.0f * sn * cs, c2 = 1.0f - 2.0f * sn * sn;
                    sn = s2; cs = c2; fr *= 2.0f;
                }
            }
#pragma unroll
            for (int f = 0; f < 39; ++f) tb[9 + f] = ta[f];
        }
        #ifdef MP_EXP_SIGCACHED
        const SigIO sio = {const_cast<char*>(sigbuf) + ((size_t)(blockIdx.x) * WAVES + wave) * (size_t)(8 * SIG_LAYER), SIG_LAYER};
#else
        const SigIO sio = {const_cast<char*>(sigbuf) + ((size_t)t * WAVES + wave) * (size_t)(8 * SIG_LAYER), SIG_LAYER};
#endif
        // V_7 = sigma'_7 (.) W_8[sdf row]
        opx8 Bcur[KS_REG][NB];
        __syncthreads();   // w8 visible
#pragma unroll
        for (int ks = 0; ks < KS_REG; ++ks) {
            const opx8 wv = *(const opx8*)(w8 + ks * 32 + 8 * gq);
            if constexpr (SIG8) {
                // chunk ks: dword 2 nb + mbl = the bytes of row pairs (mbl, j = 0, 1) of column block nb -> halves 0 .. 7 in operand order
                const u32x4 pk = *(const u32x4*)(sio.base + (size_t)7 * SIG_LAYER + ks * 1024 + lane * 16);
                const opx8 wq = wv * (op_t)SIG_QINV_F;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const unsigned d0 = nb == 0 ? pk.x : pk.z, d1 = nb == 0 ? pk.y : pk.w;
                    u32x4 h;   // halves 1024 + b
                    h.x = __builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u);
                    h.y = __builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u);
                    h.z = __builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u);
                    h.w = __builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u);
                    Bcur[ks][nb] = (__builtin_bit_cast(opx8, h) - (op_t)1024.0f) * wq;
                }
            } else {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    Bcur[ks][nb] = *(const opx8*)(sio.base + (size_t)7 * SIG_LAYER + (ks * NB + nb) * 1024 + lane * 16) * wv;
            }
        }
        float g[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        f32x4 out[NB];
#ifdef MP_EXP_GRAD_OLD
        prologue<KS_IN, WAVES, false>(net, wpack, smem, wave, lane);
#else
        prologue<KS_IN, WAVES>(net, wpack, smem, wave, lane);   // barrier inside: tables visible
#endif
        MP_STAMP_AT(HID_SIGMUL, 120, 2);
        run_net<NB, false, KS_IN, HID_SIGMUL, WAVES, GradCapture>(net, wpack, nullptr, smem, Bcur, nullptr, out, wave, lane, sio,
                                                                   GradCapture{tabs, tabs + PTS * 48, g});
        MP_STAMP_AT(HID_SIGMUL, 120, 3);
        GradCapture::unrotate(g);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                g[nb][a] += __shfl_xor(g[nb][a], 16);
                g[nb][a] += __shfl_xor(g[nb][a], 32);
            }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int pid = __shfl(id, nb * 16 + (lane & 15));
            if (lane < 16 && pid >= 0) {
                const float gx = g[nb][0], gy = g[nb][1], gz = g[nb][2];
                const float* Ji = jinv + 9 * (size_t)pid;
                float n0 = gx * Ji[0] + gy * Ji[3] + gz * Ji[6];
                float n1 = gx * Ji[1] + gy * Ji[4] + gz * Ji[7];
                float n2 = gx * Ji[2] + gy * Ji[5] + gz * Ji[8];
                float inv = 1.0f / fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-12f);  // F.normalize default eps
                n0 *= inv; n1 *= inv; n2 *= inv;
                inv = 1.0f / fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-6f);         // multiply.py:606
                normal_out[3 * (size_t)pid] = n0 * inv;
                normal_out[3 * (size_t)pid + 1] = n1 * inv;
                normal_out[3 * (size_t)pid + 2] = n2 * inv;
            }
        }
        __syncthreads();   // the tables are rebuilt by the next tile
    }
}

// ------------------------------------------------------------------------------------------------ colour
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_mlp_color(const NetDesc net, const char* __restrict__ wpack,
                                                          const float* __restrict__ bias, const float* __restrict__ xc,
                                                          const float* __restrict__ normal,
                                                          const char* __restrict__ feat_frag,
                                                          const int* __restrict__ worklist,
                                                          const int* __restrict__ count_p, int max_count,
                                                          float* __restrict__ rgb_out) {
    constexpr int KS_IN = 2;
    using L = Lds<KS_IN, NB, WAVES>;
    static_assert(64 % L::PTS == 0, "a wave consumes a whole number of column blocks of one 64-item shade tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int count = count_p ? min(*count_p, max_count) : max_count;
    float* bias_lds = (float*)(smem + L::bias0);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * in_stride(KS_IN);
    load_bias(net, bias, bias_lds);
    for (int t = blockIdx.x; t * L::TILE < count; t += gridDim.x) {
        MP_STAMP_AT(HID_RELU, 120, 0);
        const int w0 = t * L::TILE + wave * L::PTS;   // first work item of this wave
        const int tile = w0 / 64, nb0 = (w0 % 64) / 16;
        const int w = w0 + lane;
        // Order of the tile's memory requests (round 6): the work item's id first (one small load), then the first weight chunks
        // and the 16 KiB of feature fragments of this wave -- none of them depends on the id -- and only then the id -> position /
        // normal chain, whose two dependent latencies now run beside the big transfers instead of in front of them.
        const int id = (lane < L::PTS && w < count) ? (worklist ? worklist[w] : w) : -1;
        prologue_issue<KS_IN, WAVES>(net, wpack, smem + L::ring, wave, lane);
        opx8 Bcur[KS_REG][NB];
        const bool live = w0 < count;
#pragma unroll
        for (int ks = 0; ks < KS_REG; ++ks)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                Bcur[ks][nb] = live ? *(const opx8*)(feat_frag + (((size_t)tile * KS_REG + ks) * 4 + nb0 + nb) * 1024 + lane * 16)
                                    : (opx8)(op_t)0.0f;
        if (lane < L::PTS) {
            op_t* row = stage + lane * in_stride(KS_IN);
#pragma unroll
            for (int f = 0; f < KS_IN * 32; ++f) row[f] = (op_t)0.0f;
            if (id >= 0) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    row[a] = (op_t)xc[3 * (size_t)id + a];
                    row[3 + a] = (op_t)normal[3 * (size_t)id + a];
                }
            }
        }
        MP_STAMP_AT(HID_RELU, 120, 1);
        f32x4 out[NB];
        prologue_wait();
        MP_STAMP_AT(HID_RELU, 120, 2);
        run_net<NB, false, KS_IN, HID_RELU, WAVES>(net, wpack, bias_lds, smem + L::ring, Bcur, stage, out, wave, lane);
        MP_STAMP_AT(HID_RELU, 120, 3);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int pid = __shfl(id, nb * 16 + (lane & 15));
            if (lane < 16 && pid >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb_out[3 * (size_t)pid + c] = 1.0f / (1.0f + __expf(-out[nb][c]));
            }
        }
        MP_STAMP_AT(HID_RELU, 121, 0);
    }
}

// ------------------------------------------------------------------------------------------------ background
// A wave owns 16*NB consecutive samples; with NB = 2 and 32 samples per ray that is exactly one ray.
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_background(const NetDesc net_imp, const char* __restrict__ wp_imp,
                                                           const float* __restrict__ bias_imp, const NetDesc net_ren,
                                                           const char* __restrict__ wp_ren,
                                                           const float* __restrict__ bias_ren,
                                                           const float* __restrict__ dirs, const float* __restrict__ cam,
                                                           const float* __restrict__ z_bg, int z_per_ray, int n_rays,
                                                           float radius, float* __restrict__ bg_rgb) {
    constexpr int KS_IN = 3, NBG = 32;
    using L = Lds<KS_IN, NB, WAVES>;
    static_assert(L::PTS % NBG == 0, "whole rays per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* bias_lds0 = (float*)(smem + L::bias0);
    float* bias_lds1 = (float*)(smem + L::bias1);
    op_t* stage = (op_t*)(smem + L::stage) + wave * L::PTS * in_stride(KS_IN);
    float* scr = (float*)(smem + L::scratch) + wave * L::PTS * 4;
    load_bias(net_imp, bias_imp, bias_lds0);
    load_bias(net_ren, bias_ren, bias_lds1);
    const int n_pts = n_rays * NBG;
    const float ox = cam[0], oy = cam[1], oz = cam[2];
    for (int t = blockIdx.x; t * L::TILE < n_pts; t += gridDim.x) {
        prologue_issue<KS_IN, WAVES>(net_imp, wp_imp, smem + L::ring, wave, lane);
        const int q = t * L::TILE + wave * L::PTS + lane;
        const int ray = q / NBG, s = q % NBG;
        float d[3] = {0.f, 0.f, 1.f};
        if (lane < L::PTS) {
            const bool ok = ray < n_rays;
            float depth = 0.1f;
            if (ok) {
                d[0] = dirs[3 * ray]; d[1] = dirs[3 * ray + 1]; d[2] = dirs[3 * ray + 2];
                depth = z_per_ray ? z_bg[(size_t)ray * NBG + s] : z_bg[s];
            }
            // depth2pts_outside (multiply.py:698-726)
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
#pragma unroll
            for (int a = 0; a < 3; ++a) pn[a] = ps[a] * ca + cr[a] * sa + ax[a] * adp * (1.0f - ca);
            const float pnn = sqrtf(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]);
            const float x4[4] = {pn[0] / pnn, pn[1] / pnn, pn[2] / pnn, depth};
            stage_pe<4, 10, KS_IN>(stage + lane * in_stride(KS_IN), x4);
        }
        opx8 Bcur[KS_REG][NB];
        f32x4 out[NB];
        zero_b<NB>(Bcur);
        prologue_wait();
        run_net<NB, false, KS_IN, HID_SOFTPLUS, WAVES>(net_imp, wp_imp, bias_lds0, smem + L::ring, Bcur, stage, out, wave,
                                                       lane);
        // the colour net's first chunks: behind the last barrier of the network above the ring is free; their latency runs beside
        // the density / view-direction staging below
        prologue_issue<KS_IN, WAVES>(net_ren, wp_ren, smem + L::ring, wave, lane);
        if (lane < 16) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) scr[(nb * 16 + lane) * 4 + 3] = fabsf(out[nb][0]);  // AbsDensity (density.py:32-34)
        }
        // colour net: [PE_4(view dir) (27), frame code (hoisted), features (registers)]
        if (lane < L::PTS) stage_pe<3, 4, KS_IN>(stage + lane * in_stride(KS_IN), d);
        prologue_wait();
        run_net<NB, false, KS_IN, HID_RELU, WAVES>(net_ren, wp_ren, bias_lds1, smem + L::ring, Bcur, stage, out, wave, lane);
        if (lane < 16) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int c = 0; c < 3; ++c) scr[(nb * 16 + lane) * 4 + c] = 1.0f / (1.0f + __expf(-out[nb][c]));
        }
        __syncthreads();
        // bg_volume_rendering (multiply.py:682-696): the first lanes composite one ray each
        if (lane < L::PTS / NBG) {
            const int r = (t * L::TILE + wave * L::PTS) / NBG + lane;
            if (r < n_rays) {
                float acc[3] = {0.f, 0.f, 0.f}, csum = 0.0f;
                for (int i = 0; i < NBG; ++i) {
                    const float zi = z_per_ray ? z_bg[(size_t)r * NBG + i] : z_bg[i];
                    const float zn = i + 1 < NBG ? (z_per_ray ? z_bg[(size_t)r * NBG + i + 1] : z_bg[i + 1]) : 0.f;
                    const float dist = i + 1 < NBG ? zi - zn : 1e10f;
                    const float* sp = scr + (lane * NBG + i) * 4;
                    const float fe = dist * sp[3];
                    const float alpha = 1.0f - expf(-fe);
                    const float wgt = alpha * expf(-csum);
                    acc[0] += wgt * sp[0]; acc[1] += wgt * sp[1]; acc[2] += wgt * sp[2];
                    csum += fe;
                }
                bg_rgb[3 * r] = acc[0]; bg_rgb[3 * r + 1] = acc[1]; bg_rgb[3 * r + 2] = acc[2];
            }
        }
    }
}

// persistent grid: `per_cu` workgroups per CU, grid-stride over tiles
int grid_for(int work_blocks, int per_cu) {
    const int cap = 256 * per_cu;
    return work_blocks < cap ? (work_blocks > 0 ? work_blocks : 1) : cap;
}

NetDesc as_desc(const MpNet* net) {
    NetDesc d;
    __builtin_memcpy(&d, net, sizeof(d));
    return d;
}

// The core runs a network as a run of hidden (activated) layers followed by its linear output layer(s)
// (mlp_core.hpp run_net); anything else is a malformed descriptor.
bool net_ok(const MpNet* net) {
    if (!net || net->n_layers < 1 || net->n_layers > MAX_LAYERS) return false;
    bool linear_seen = false;
    int chunks = 0;
    for (int l = 0; l < net->n_layers; ++l) {
        const auto& L = net->layer[l];
        if (L.n_chunk < 1 || L.n_chunk > MAX_CHUNKS) return false;
        if (L.act == ACT_NONE) linear_seen = true;
        else if (linear_seen) return false;
        chunks += L.n_chunk;
    }
    return chunks == net->total_chunks;
}

constexpr int PNB = 2, PWAVES = 8;   // plain-mode geometry

}  // namespace

#ifdef MP_EXP_STAMP
extern "C" int mp_debug_stamps(void* dst_host) {
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(mp::mp_stamps), sizeof(mp::mp_stamps));
}
#endif

extern "C" int mp_mlp_sdf(const MpNet* net, const void* wpack, const float* bias, const float* xc,
                          const int* worklist, const int* count, int max_count, float* sdf_out, void* stream) {
    if (max_count <= 0) return 0;
    if (!net_ok(net)) return -1;
    hipStream_t st = (hipStream_t)stream;
    using L = Lds<2, PNB, PWAVES>;
    MP_LDS_ATTR((k_mlp_sdf<PNB, PWAVES>), L::total);
    const NetDesc d = as_desc(net);
    hipLaunchKernelGGL((k_mlp_sdf<PNB, PWAVES>), dim3(grid_for((max_count + L::TILE - 1) / L::TILE, 1)),
                       dim3(PWAVES * 64), L::total, st, d, (const char*)wpack, bias, xc, worklist, count, max_count,
                       sdf_out);
    return (int)hipGetLastError();
}

extern "C" int mp_mlp_sdf_x2(const MpNet* net, const void* wpack, const float* bias, const float* xc,
                             const int* worklist, const int* count, int max_count, float* sdf_out, void* stream) {
    if (max_count <= 0) return 0;
    if (!net_ok(net)) return -1;
    hipStream_t st = (hipStream_t)stream;
    using L = Lds<2, PNB, PWAVES>;
    constexpr int TILE = 16 * PWAVES;
    MP_LDS_ATTR((k_mlp_sdf_x2<PWAVES>), L::total);
    const NetDesc d = as_desc(net);
    hipLaunchKernelGGL((k_mlp_sdf_x2<PWAVES>), dim3(grid_for((max_count + TILE - 1) / TILE, 1)), dim3(PWAVES * 64), L::total, st,
                       d, (const char*)wpack, bias, xc, worklist, count, max_count, sdf_out);
    return (int)hipGetLastError();
}

extern "C" int mp_mlp_full(const MpNet* net, const void* wpack, const float* bias, const float* x, int d_in, int n,
                           float* out, void* stream) {
    if (n <= 0) return 0;
    if (!net_ok(net)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const NetDesc d = as_desc(net);
    if (d_in == 3) {
        using L = Lds<2, PNB, PWAVES>;
        MP_LDS_ATTR((k_mlp_full<3, 6, 2, PNB, PWAVES>), L::total);
        hipLaunchKernelGGL((k_mlp_full<3, 6, 2, PNB, PWAVES>), dim3(grid_for((n + L::TILE - 1) / L::TILE, 1)),
                           dim3(PWAVES * 64), L::total, st, d, (const char*)wpack, bias, x, n, out);
    } else if (d_in == 4) {
        using L = Lds<3, PNB, PWAVES>;
        MP_LDS_ATTR((k_mlp_full<4, 10, 3, PNB, PWAVES>), L::total);
        hipLaunchKernelGGL((k_mlp_full<4, 10, 3, PNB, PWAVES>), dim3(grid_for((n + L::TILE - 1) / L::TILE, 1)),
                           dim3(PWAVES * 64), L::total, st, d, (const char*)wpack, bias, x, n, out);
    } else {
        return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int mp_sig_bytes_per_point(void) { return 8 * KS_REG * SIG_CHUNK_BYTES / 32; }

extern "C" int mp_mlp_shade_rev(const MpNet* net, const void* wpack, const float* bias, const MpNet* gnet, const void* gpack,
                                const void* w8_slots, const float* xc, const float* jinv, const int* worklist,
                                const int* count, int max_count, float* sdf_out, float* normal_out, void* feat_frag,
                                void* sig, int seg_points, void* stream) {
    if (max_count <= 0) return 0;
    if (seg_points < 256 || seg_points % 256) return -1;
    if (!net_ok(net) || !net_ok(gnet)) return -1;
    // the reverse sweep fetches its stored sigmoids one chunk ahead into alternating buffers (mlp_core.hpp run_layer_pp):
    // every sigmoid-multiplied layer must be 8 chunks (256 rows) so that the alternation carries across layers
    for (int l = 0; l < gnet->n_layers; ++l)
        if (gnet->layer[l].act == ACT_SIGMUL && gnet->layer[l].n_chunk != KS_REG) return -1;
    hipStream_t st = (hipStream_t)stream;
    using L = Lds<2, PNB, PWAVES>;
    constexpr int TILE = 16 * PNB * PWAVES;
    constexpr int LDS_G = RING_SLOTS * chunk_bytes(0) + 512 + PWAVES * 2 * 16 * PNB * 48 * 2;
    MP_LDS_ATTR((k_mlp_fwdsave<PNB, PWAVES>), L::total);
    MP_LDS_ATTR((k_mlp_grad<PNB, PWAVES>), LDS_G);
    const NetDesc d = as_desc(net), gd = as_desc(gnet);
    for (int off = 0; off < max_count; off += seg_points) {   // segments past the device-side count return at once
        const int n = max_count - off < seg_points ? max_count - off : seg_points;
        const int grid = grid_for((n + TILE - 1) / TILE, 1);
        hipLaunchKernelGGL((k_mlp_fwdsave<PNB, PWAVES>), dim3(grid), dim3(PWAVES * 64), L::total, st, d, (const char*)wpack, bias,
                           xc, worklist, count, max_count, off, seg_points, sdf_out, (char*)feat_frag, (char*)sig);
        hipLaunchKernelGGL((k_mlp_grad<PNB, PWAVES>), dim3(grid), dim3(PWAVES * 64), LDS_G, st, gd, (const char*)gpack,
                           (const op_t*)w8_slots, xc, jinv, worklist, count, max_count, off, seg_points, (const char*)sig,
                           normal_out);
    }
    return (int)hipGetLastError();
}

extern "C" int mp_mlp_shade(const MpNet* net, const void* wpack, const float* bias, const float* xc,
                            const float* jinv, const int* worklist, const int* count, int max_count, float* sdf_out,
                            float* normal_out, void* feat_frag, void* stream) {
    if (max_count <= 0) return 0;
    if (!net_ok(net)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const NetDesc d = as_desc(net);
    using L = Lds<2, 2, 8>;
    MP_LDS_ATTR((k_mlp_shade), L::total);
    hipLaunchKernelGGL(k_mlp_shade, dim3(grid_for((max_count + 63) / 64, 1)), dim3(512), L::total, st, d,
                       (const char*)wpack, bias, xc, jinv, worklist, count, max_count, sdf_out, normal_out,
                       (char*)feat_frag);
    return (int)hipGetLastError();
}

extern "C" int mp_mlp_color(const MpNet* net, const void* wpack, const float* bias, const float* xc,
                            const float* normal, const void* feat_frag, const int* worklist, const int* count,
                            int max_count, float* rgb_out, void* stream) {
    if (max_count <= 0) return 0;
    if (!net_ok(net)) return -1;
    hipStream_t st = (hipStream_t)stream;
    using L = Lds<2, PNB, PWAVES>;
    MP_LDS_ATTR((k_mlp_color<PNB, PWAVES>), L::total);
    const NetDesc d = as_desc(net);
    hipLaunchKernelGGL((k_mlp_color<PNB, PWAVES>), dim3(grid_for((max_count + L::TILE - 1) / L::TILE, 1)),
                       dim3(PWAVES * 64), L::total, st, d, (const char*)wpack, bias, xc, normal, (const char*)feat_frag,
                       worklist, count, max_count, rgb_out);
    return (int)hipGetLastError();
}

extern "C" int mp_background(const MpNet* net_imp, const void* wpack_imp, const float* bias_imp, const MpNet* net_ren,
                             const void* wpack_ren, const float* bias_ren, const float* dirs, const float* cam,
                             const float* z_bg, int z_per_ray, int n_rays, float radius, float* bg_rgb, void* stream) {
    if (n_rays <= 0) return 0;
    if (!net_ok(net_imp) || !net_ok(net_ren)) return -1;
    hipStream_t st = (hipStream_t)stream;
    using L = Lds<3, PNB, PWAVES>;
    MP_LDS_ATTR((k_background<PNB, PWAVES>), L::total);
    const NetDesc d0 = as_desc(net_imp), d1 = as_desc(net_ren);
    hipLaunchKernelGGL((k_background<PNB, PWAVES>), dim3(grid_for((n_rays * 32 + L::TILE - 1) / L::TILE, 1)),
                       dim3(PWAVES * 64), L::total, st, d0, (const char*)wpack_imp, bias_imp, d1, (const char*)wpack_ren,
                       bias_ren, dirs, cam, z_bg, z_per_ray, n_rays, radius, bg_rgb);
    return (int)hipGetLastError();
}
