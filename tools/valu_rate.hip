// VALU instruction throughput on one SIMD (standalone):  hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/valu_rate.hip
// Each wave issues ITER x 32 independent instances of ONE instruction (8 registers, round robin); 1, 2 and 4 waves per SIMD.
// cycles per instruction per SIMD = wall clock x (measured shader clock) / instructions issued on that SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(INS) for (int it = 0; it < iters; ++it) { REP8(INS) REP8(INS) REP8(INS) REP8(INS) }

#define I_PKADD(i) asm volatile("v_pk_add_f16 %0, %0, 1.0 op_sel_hi:[1,0]" : "+v"(r[i]));
#define I_PKFMA(i) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(r[i]));
#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
#define I_EXP16(i) asm volatile("v_exp_f16_e32 %0, %0" : "+v"(r[i]));
#define I_EXP16S(i) asm volatile("v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r[i]));
#define I_EXP16SL(i) asm volatile("v_exp_f16_sdwa %0, -|%0| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(r[i]));
#define I_LOG16S(i) asm volatile("v_log_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r[i]));
#define I_EXP32(i) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(r[i]));
#define I_RCP16(i) asm volatile("v_rcp_f16_e32 %0, %0" : "+v"(r[i]));
#define I_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(r[i]));
#define I_MAXI16(i) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(r[i]));
#define I_MOV(i) asm volatile("v_mov_b32 %0, %0" : "+v"(r[i]));
#define I_NOP(i) asm volatile("s_nop 0");
#define I_PKMUL(i) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(r[i]));
#define I_PERM(i) asm volatile("v_perm_b32 %0, %0, %0, %0" : "+v"(r[i]));

#define KERNEL(NAME, INS)                                                                        \
    __global__ __launch_bounds__(1024) void NAME(int iters, unsigned long long* cyc, unsigned* sink) { \
        unsigned r[8];                                                                           \
        for (int i = 0; i < 8; ++i) r[i] = 0x38003800u + threadIdx.x + i;                        \
        __syncthreads();                                                                         \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        BODY(INS)                                                                                \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned s = 0;                                                                          \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                                   \
        if (s == 0x12345u) sink[0] = s;                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;                               \
    }
KERNEL(k_pkadd, I_PKADD) KERNEL(k_pkfma, I_PKFMA) KERNEL(k_fma32, I_FMA32) KERNEL(k_exp16, I_EXP16) KERNEL(k_exp16s, I_EXP16S)
KERNEL(k_exp16sl, I_EXP16SL) KERNEL(k_log16s, I_LOG16S) KERNEL(k_exp32, I_EXP32) KERNEL(k_rcp16, I_RCP16) KERNEL(k_cvtpk, I_CVTPK)
KERNEL(k_maxi16, I_MAXI16) KERNEL(k_mov, I_MOV) KERNEL(k_nop, I_NOP) KERNEL(k_pkmul, I_PKMUL) KERNEL(k_perm, I_PERM)

template <typename K>
void run(const char* name, K kern) {
    const int iters = 2000;
    unsigned long long* cyc; unsigned* sink;
    (void)hipMalloc(&cyc, 8); (void)hipMalloc(&sink, 4);
    printf("%-34s", name);
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, 10, cyc, sink);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, iters, cyc, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double per_wave = (double)c / (iters * 32.0);                  // wave 0's own cycles per instruction
        const double ns_simd = ms * 1e6 / (iters * 32.0 * (threads / 256));  // wall ns per instruction issued on a SIMD
        printf("  %d w/SIMD: %5.2f cyc/instr (wave 0) %5.2f ns/instr/SIMD |", threads / 256, per_wave, ns_simd);
    }
    printf("\n");
}
int main() {
    run("v_pk_add_f16", k_pkadd); run("v_pk_mul_f16", k_pkmul); run("v_pk_fma_f16", k_pkfma); run("v_fma_f32", k_fma32);
    run("v_pk_max_i16", k_maxi16); run("v_cvt_pk_f16_f32", k_cvtpk); run("v_mov_b32", k_mov); run("v_perm_b32", k_perm); run("s_nop 0", k_nop);
    run("v_exp_f16_sdwa -|x| DWORD pad", k_exp16sl); run("v_log_f16_sdwa WORD_1 preserve", k_log16s);
    run("v_exp_f32", k_exp32); run("v_rcp_f16", k_rcp16);
    return 0;
}
