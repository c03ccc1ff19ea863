// Round 6 variant of pair_model.hip: does the VALU slow-down beside a running MFMA stream depend on WHERE the MFMA's operands live?
// Waves 0..3 run MFMAs 16x16x32 f16 whose accumulators (and, second variant, B operands; third, A and B) sit in AGPRs -- the
// accumulator half of the unified register file -- while waves 4..7 run packed-half / transcendental VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define I_PK(i) asm volatile("v_pk_add_f16 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(r[i]));
#define I_EXP(i) asm volatile("v_exp_f16_sdwa %0, -|%0| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(r[i]));

// OPS: 0 = everything in VGPRs (compiler's choice, builtin), 1 = accumulators in AGPRs, 2 = accumulators + B in AGPRs, 3 = acc + A + B in AGPRs
template <int VK, int OPS>
__global__ __launch_bounds__(512) void k_pair(int iters, int do_m, int do_v, unsigned long long* cyc, float* sink) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (wave < 4) {
        if (do_m) {
            h8 a, b0, b1;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b0[i] = (_Float16)(0.002f * i); b1[i] = (_Float16)(0.003f * i); }
            f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            if constexpr (OPS == 0) {
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, c2, 0, 0, 0);
                        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, c3, 0, 0, 0);
                    }
                }
            } else {
                f4 ca0, ca1, ca2, ca3;
                h8 ab0, ab1, aa;
                asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(ca0[0])); 
#define ZACC(c) asm volatile("v_accvgpr_write_b32 %0, 0\n\tv_accvgpr_write_b32 %1, 0\n\tv_accvgpr_write_b32 %2, 0\n\tv_accvgpr_write_b32 %3, 0" : "=a"(c[0]), "=a"(c[1]), "=a"(c[2]), "=a"(c[3]));
                f4 z = {0, 0, 0, 0};
                asm volatile("" : "=a"(ca0) : "0"(z));
                asm volatile("" : "=a"(ca1) : "0"(z));
                asm volatile("" : "=a"(ca2) : "0"(z));
                asm volatile("" : "=a"(ca3) : "0"(z));
                asm volatile("" : "=a"(ab0) : "0"(b0));
                asm volatile("" : "=a"(ab1) : "0"(b1));
                asm volatile("" : "=a"(aa) : "0"(a));
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if constexpr (OPS == 1) {
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca0) : "v"(a), "v"(b0));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca1) : "v"(a), "v"(b1));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca2) : "v"(a), "v"(b0));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca3) : "v"(a), "v"(b1));
                        } else if constexpr (OPS == 2) {
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca0) : "v"(a), "a"(ab0));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca1) : "v"(a), "a"(ab1));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca2) : "v"(a), "a"(ab0));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca3) : "v"(a), "a"(ab1));
                        } else {
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca0) : "a"(aa), "a"(ab0));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca1) : "a"(aa), "a"(ab1));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca2) : "a"(aa), "a"(ab0));
                            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ca3) : "a"(aa), "a"(ab1));
                        }
                    }
                }
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                asm volatile("" : "=v"(c0) : "0"(ca0));
                asm volatile("" : "=v"(c1) : "0"(ca1));
                asm volatile("" : "=v"(c2) : "0"(ca2));
                asm volatile("" : "=v"(c3) : "0"(ca3));
            }
            s = c0[0] + c1[0] + c2[0] + c3[0];
        }
    } else {
        if (do_v) {
            unsigned r[8];
            for (int i = 0; i < 8; ++i) r[i] = 0x38003800u + lane + i;
            for (int it = 0; it < iters; ++it) {
                if constexpr (VK == 0) { REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) REP8(I_PK) }
                if constexpr (VK == 1) { REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) REP8(I_EXP) }
            }
            for (int i = 0; i < 8; ++i) s += (float)r[i];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern) {
    const int iters = 4000;
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, 64); (void)hipMalloc(&sink, 4);
    printf("%-44s", name);
    for (int mode = 0; mode < 3; ++mode) {
        const int dm = mode != 1, dv = mode != 0;
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, 10, dm, dv, cyc, sink);
        (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, iters, dm, dv, cyc, sink);
        (void)hipDeviceSynchronize();
        unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf(" | %s: %6.2f cyc/MFMA %5.2f cyc/VALU", mode == 0 ? "M alone" : mode == 1 ? "V alone" : "both   ",
               (double)c[0] / (iters * 32.0), (double)c[4] / (iters * 64.0));
    }
    printf("\n");
}
int main() {
    run("v_pk_add_f16 | all operands VGPR", k_pair<0, 0>);
    run("v_pk_add_f16 | acc AGPR", k_pair<0, 1>);
    run("v_pk_add_f16 | acc + B AGPR", k_pair<0, 2>);
    run("v_pk_add_f16 | acc + A + B AGPR", k_pair<0, 3>);
    run("v_exp_f16 | all operands VGPR", k_pair<1, 0>);
    run("v_exp_f16 | acc AGPR", k_pair<1, 1>);
    run("v_exp_f16 | acc + B AGPR", k_pair<1, 2>);
    run("v_exp_f16 | acc + A + B AGPR", k_pair<1, 3>);
    return 0;
}
