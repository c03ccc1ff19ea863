// Machine model of one CU for the fused-MLP K-step stream (standalone; no product code):
//   hipcc --offload-arch=gfx950 -O3 -o issue_model tools/issue_model.hip && ./issue_model
// Each wave runs ITER "blocks" of 8 K steps; a K step = 2 MFMA 16x16x32 f16 (two accumulators) + a filler pattern.
// Reports shader cycles per block (s_memtime, wave 0 of workgroup 0) and wall time, for 1 and 2 waves per SIMD, so that
// the cost of activation VALU work between MFMAs can be read off directly (MFMA floor = 16 MFMAs x 16 cycles = 256).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define SB __builtin_amdgcn_sched_barrier(0)
#define PKADD(r) asm volatile("v_pk_add_f16 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(r))
#define EXPF(r) asm volatile("v_exp_f16_e32 %0, %0" : "+v"(r))
#define EXP2(d, s)                                                                                             \
    asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 0\n\t"   \
                 "v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"             \
                 : "=&v"(d) : "v"(s))
#define EXP2NA(d, s)                                                                                              \
    asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 0\n\t" \
                 "v_exp_f16_sdwa %0, -|%1| dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"           \
                 : "=&v"(d) : "v"(s))
#define LOG2(d, s)                                                                                             \
    asm volatile("v_log_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\ts_nop 0\n\t"   \
                 "v_log_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1"             \
                 : "=&v"(d) : "v"(s))
#define ADD1(d, s) asm volatile("s_nop 0\n\tv_pk_add_f16 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(s))
#define ADD(d, a, b) asm volatile("s_nop 0\n\tv_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define SUB(d, a, b) asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b))
#define RELU(d, s) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(d) : "v"(s))
#define CVT(d, a, b) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))

// serial chain of one row pair (what mlp_core.hpp's softplus+sigmoid piece issues)
#define CHAIN(z, out_h, out_s)        \
    {                                 \
        unsigned u, w, lg, r, h, d;   \
        EXP2NA(u, z);                 \
        ADD1(w, u);                   \
        LOG2(lg, w);                  \
        RELU(r, z);                   \
        ADD(h, r, lg);                \
        SUB(d, z, h);                 \
        EXP2(out_s, d);               \
        out_h = h;                    \
    }

template <int MODE>
__global__ __launch_bounds__(512) void k_model(int iters, unsigned long long* cyc, float* sink, int store, uint4* gbuf) {
    const int lane = threadIdx.x & 63;
    h8 a, b0, b1;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b0[i] = (_Float16)(0.002f * i); b1[i] = (_Float16)(0.003f * i); }
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    unsigned f[16];
    for (int i = 0; i < 16; ++i) f[i] = 0x3c003800u + lane + i;   // half pairs around (1.0, 0.5)
    unsigned hs[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc1, 0, 0, 0);
            SB;
            if constexpr (MODE == 1) { PKADD(f[0]); PKADD(f[1]); PKADD(f[2]); }                                  // 3 plain / K step
            if constexpr (MODE == 2) { PKADD(f[0]); PKADD(f[1]); PKADD(f[2]); PKADD(f[3]); PKADD(f[4]); PKADD(f[5]); }   // 6 plain
            if constexpr (MODE == 3) { EXPF(f[0]); EXPF(f[1]); EXPF(f[2]); }                                      // 3 trans
            if constexpr (MODE == 4) { EXPF(f[0]); EXPF(f[1]); EXPF(f[2]); EXPF(f[3]); EXPF(f[4]); EXPF(f[5]); } // 6 trans
            if constexpr (MODE == 5) { PKADD(f[0]); EXPF(f[1]); PKADD(f[2]); EXPF(f[3]); PKADD(f[4]); EXPF(f[5]); }   // 3 + 3
            if constexpr (MODE == 6) {   // 12 plain
                PKADD(f[0]); PKADD(f[1]); PKADD(f[2]); PKADD(f[3]); PKADD(f[4]); PKADD(f[5]);
                PKADD(f[6]); PKADD(f[7]); PKADD(f[8]); PKADD(f[9]); PKADD(f[10]); PKADD(f[11]);
            }
            if constexpr (MODE == 7) {   // the real thing: serial chains in K steps 0..3 (as mlp_core.hpp today)
                if (ks < 4) { unsigned z; CVT(z, f[2 * ks], f[2 * ks + 1]); CHAIN(z, hs[ks], ss[ks]); }
            }
            SB;
        }
        if constexpr (MODE == 7) {
            if (store) {   // two 16 B / lane stores per block, streaming through a 2 GiB buffer like the sigmoid fragments
                uint4* g = gbuf + ((size_t)(it & 127) * 256 + blockIdx.x) * 1024 + threadIdx.x;
                g[0] = make_uint4(ss[0], ss[1], ss[2], ss[3]);
                g[512] = make_uint4(hs[0], hs[1], hs[2], hs[3]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc1[0];
    for (int i = 0; i < 16; ++i) s += (float)f[i];
    for (int i = 0; i < 4; ++i) s += (float)hs[i] + (float)ss[i];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// lock-step variant (its own kernel: needs named temporaries across K steps)
__global__ __launch_bounds__(512) void k_lock(int iters, unsigned long long* cyc, float* sink, int store, uint4* gbuf) {
    const int lane = threadIdx.x & 63;
    h8 a, b0, b1;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b0[i] = (_Float16)(0.002f * i); b1[i] = (_Float16)(0.003f * i); }
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    unsigned f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0x3c003800u + lane + i;
    unsigned z[4], u[4], w[4], lg[4], r[4], h[4] = {0, 0, 0, 0}, d[4], s4[4] = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define MF                                                                  \
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc0, 0, 0, 0);    \
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc1, 0, 0, 0);    \
    SB;
#define EXP2NA_(d, s) asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(d) : "v"(s))
#define EXP2NA_H(d, s) asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(d) : "v"(s))
#define LOG_(d, s) asm volatile("v_log_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(d) : "v"(s))
#define LOG_H(d, s) asm volatile("v_log_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(d) : "v"(s))
#define EXP_(d, s) asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(d) : "v"(s))
#define EXP_H(d, s) asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(d) : "v"(s))
#define ADD1_(d, s) asm volatile("v_pk_add_f16 %0, %1, 1.0 op_sel_hi:[1,0]" : "=v"(d) : "v"(s))
#define ADD_(d, a, b) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
    for (int it = 0; it < iters; ++it) {
        // 48 VALU instructions = 6 per K step; every instruction's inputs were produced >= 4 instructions earlier
        MF CVT(z[0], f[0], f[1]); CVT(z[1], f[2], f[3]); CVT(z[2], f[4], f[5]); CVT(z[3], f[6], f[7]); EXP2NA_(u[0], z[0]); EXP2NA_(u[1], z[1]); SB;
        MF EXP2NA_(u[2], z[2]); EXP2NA_(u[3], z[3]); EXP2NA_H(u[0], z[0]); EXP2NA_H(u[1], z[1]); EXP2NA_H(u[2], z[2]); EXP2NA_H(u[3], z[3]); SB;
        MF RELU(r[0], z[0]); RELU(r[1], z[1]); ADD1_(w[0], u[0]); ADD1_(w[1], u[1]); ADD1_(w[2], u[2]); ADD1_(w[3], u[3]); SB;
        MF RELU(r[2], z[2]); RELU(r[3], z[3]); LOG_(lg[0], w[0]); LOG_(lg[1], w[1]); LOG_(lg[2], w[2]); LOG_(lg[3], w[3]); SB;
        MF LOG_H(lg[0], w[0]); LOG_H(lg[1], w[1]); LOG_H(lg[2], w[2]); LOG_H(lg[3], w[3]); ADD_(h[0], r[0], lg[0]); ADD_(h[1], r[1], lg[1]); SB;
        MF ADD_(h[2], r[2], lg[2]); ADD_(h[3], r[3], lg[3]); SUB(d[0], z[0], h[0]); SUB(d[1], z[1], h[1]); SUB(d[2], z[2], h[2]); SUB(d[3], z[3], h[3]); SB;
        MF EXP_(s4[0], d[0]); EXP_(s4[1], d[1]); EXP_(s4[2], d[2]); EXP_(s4[3], d[3]); EXP_H(s4[0], d[0]); EXP_H(s4[1], d[1]); SB;
        MF EXP_H(s4[2], d[2]); EXP_H(s4[3], d[3]); SB;
        if (store) {
            uint4* g = gbuf + ((size_t)(it & 127) * 256 + blockIdx.x) * 1024 + threadIdx.x;
            g[0] = make_uint4(s4[0], s4[1], s4[2], s4[3]);
            g[512] = make_uint4(h[0], h[1], h[2], h[3]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc1[0];
    for (int i = 0; i < 4; ++i) s += (float)h[i] + (float)s4[i];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int threads, int store) {
    const int iters = 4000;
    unsigned long long* cyc;
    float* sink;
    uint4* gbuf;
    hipMalloc(&cyc, 8);
    hipMalloc(&sink, 4);
    hipMalloc(&gbuf, (size_t)128 * 256 * 1024 * sizeof(uint4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, 10, cyc, sink, store, gbuf);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, iters, cyc, sink, store, gbuf);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_block = (double)c / iters;
    const double mfma_cyc = 256.0 * (threads / 256);   // MFMA pipe cycles per SIMD per block round (all its waves)
    printf("%-44s waves/SIMD %d store %d : %7.1f cyc/block/wave  (MFMA floor %4.0f/SIMD -> util %.2f)  wall %.3f ms  clock %.2f GHz\n",
           name, threads / 256, store, per_block, mfma_cyc, mfma_cyc / per_block, ms, c / (ms * 1e6));
    hipFree(cyc);
    hipFree(sink);
    hipFree(gbuf);
}

int main() {
    for (int threads : {256, 512}) {
        run("0 MFMA only", k_model<0>, threads, 0);
        run("1 +3 v_pk_add / K step", k_model<1>, threads, 0);
        run("2 +6 v_pk_add / K step", k_model<2>, threads, 0);
        run("6 +12 v_pk_add / K step", k_model<6>, threads, 0);
        run("3 +3 v_exp_f16 / K step", k_model<3>, threads, 0);
        run("4 +6 v_exp_f16 / K step", k_model<4>, threads, 0);
        run("5 +3 v_pk_add +3 v_exp / K step", k_model<5>, threads, 0);
        run("7 softplus+sigmoid serial chains (K 0..3)", k_model<7>, threads, 0);
        run("7 ... + one 16 B/lane store per block", k_model<7>, threads, 1);
        run("8 same, 4 chains lock-step over 8 K steps", k_lock, threads, 0);
        run("8 ... + one 16 B/lane store per block", k_lock, threads, 1);
    }
    return 0;
}
