// Two waves on one SIMD, one issuing only MFMAs, the other only VALU work: do the matrix pipe and the vector pipe of a
// SIMD run concurrently?  (standalone; hipcc --offload-arch=gfx950 -O3 -o pair_model tools/pair_model.hip)
// Workgroup = 8 waves; waves 0..3 (one per SIMD) run ITER x 32 MFMAs on 4 accumulators, waves 4..7 run ITER x 64 VALU
// instructions of one kind.  Reported: cycles per MFMA / per VALU instruction, alone and beside each other.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define I_PK(i) asm volatile("v_pk_add_f16 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(r[i]));
#define I_EXP(i) asm volatile("v_exp_f16_sdwa %0, -|%0| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(r[i]));
#define I_F32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));

template <int VK, int MK, int SWAP = 0>
__global__ __launch_bounds__(512) void k_pair(int iters, int do_m, int do_v, unsigned long long* cyc, float* sink) {
    const int wave_hw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wave = SWAP ? (wave_hw ^ 4) : wave_hw;   // SWAP: the OLDER waves (0..3) do the VALU work, the younger the MFMAs
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (wave < 4) {
        if (do_m) {
            h8 a, b0, b1;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b0[i] = (_Float16)(0.002f * i); b1[i] = (_Float16)(0.003f * i); }
            if constexpr (MK == 0) {
                f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, c2, 0, 0, 0);
                        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, c3, 0, 0, 0);
                    }
                }
                s = c0[0] + c1[0] + c2[0] + c3[0];
            } else {   // 32x32x16: 16 per iteration = the same FLOPs as 32 of the small one
                f16v c0 = {}, c1 = {};
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b0, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1, c1, 0, 0, 0);
                    }
                }
                s = c0[0] + c1[0];
            }
        }
    } else {
        if (do_v) {
            unsigned r[8];
            for (int i = 0; i < 8; ++i) r[i] = 0x38003800u + lane + i;
            for (int it = 0; it < iters; ++it) {
                if constexpr (VK == 0) { REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) }
                if constexpr (VK == 1) { REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) }
                if constexpr (VK == 2) { REP8(I_F32) REP8(I_F32) REP8(I_F32) REP8(I_F32) REP8(I_F32) REP8(I_F32) REP8(I_F32) REP8(I_F32) }
            }
            for (int i = 0; i < 8; ++i) s += (float)r[i];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int mf_per_iter) {
    const int iters = 4000;
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, 64); (void)hipMalloc(&sink, 4);
    printf("%-34s", name);
    for (int mode = 0; mode < 3; ++mode) {   // MFMA alone, VALU alone, both
        const int dm = mode != 1, dv = mode != 0;
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, 10, dm, dv, cyc, sink);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, dm, dv, cyc, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf(" | %s: %6.2f cyc/MFMA %5.2f cyc/VALU  %.3f ms", mode == 0 ? "M alone" : mode == 1 ? "V alone" : "both   ",
               (double)c[0] / (iters * (double)mf_per_iter), (double)c[4] / (iters * 64.0), ms);
    }
    printf("\n");
}
int main() {
    run("16x16x32 beside v_pk_add_f16", k_pair<0, 0>, 32);
    run("16x16x32 beside v_exp_f16_sdwa", k_pair<1, 0>, 32);
    run("16x16x32 beside v_fma_f32", k_pair<2, 0>, 32);
    printf("-- roles swapped: VALU work in the older wave of each SIMD, MFMAs in the younger\n");
    run("16x16x32 beside v_pk_add_f16 (swap)", k_pair<0, 0, 1>, 32);
    run("16x16x32 beside v_exp_f16_sdwa (swap)", k_pair<1, 0, 1>, 32);
    run("32x32x16 beside v_pk_add_f16", k_pair<0, 1>, 16);
    run("32x32x16 beside v_exp_f16_sdwa", k_pair<1, 1>, 16);
    run("32x32x16 beside v_fma_f32", k_pair<2, 1>, 16);
    return 0;
}
