// The M phase of the phase-separated MLP stream in isolation (standalone): per chunk 16 A tiles (1 KiB each) from LDS, two
// MFMA 16x16x32 per tile (column blocks), 4 accumulators, B operands from 16 register quads.  Knobs: LDS reads on/off,
// B registers distinct or shared, waves per workgroup (4 = one per SIMD, 8 = two), MFMA shape.
//   hipcc --offload-arch=gfx950 -O3 -o mphase_model tools/mphase_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int LDS, int BROT, int PF, int BIG>
__global__ __launch_bounds__(512) void k_m(int chunks, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * 16384 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 255);
    h8 B[8][2];
    for (int k = 0; k < 8; ++k) for (int nb = 0; nb < 2; ++nb) for (int e = 0; e < 8; ++e) B[k][nb][e] = (_Float16)(0.01f * (k + nb + e + lane));
    __syncthreads();
    constexpr int QN = PF + 1;
    h8 aq[QN];
    f4 acc[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
    f16v accb[2] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        const char* slot = smem + (c % 3) * 16384 + lane * 16;
#pragma unroll
        for (int t = 0; t < PF; ++t) aq[t % QN] = LDS ? *(const h8*)(slot + t * 1024) : B[t % 8][0];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const h8 a = aq[t % QN];
            if (t + PF < 16) aq[(t + PF) % QN] = LDS ? *(const h8*)(slot + (t + PF) * 1024) : B[(t + PF) % 8][1];
            __builtin_amdgcn_sched_barrier(0);
            const int ks = BROT ? t / 2 : 0, mbl = t % 2;
            if (BIG) {
                accb[mbl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[ks][0], accb[mbl], 0, 0, 0);
            } else {
                acc[mbl][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, B[ks][0], acc[mbl][0], 0, 0, 0);
                acc[mbl][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, B[ks][1], acc[mbl][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc[0][0][0] + acc[0][1][0] + acc[1][0][0] + acc[1][1][0] + accb[0][0] + accb[1][0];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}
// Round 5: (a) the A fragments read from LDS straight into ACCUMULATOR registers (ds_read_b128 a[..]) and fed to the MFMA from there
// -- does the LDS -> register write-back then stop stretching the M phase?  (b) NB = 4 column blocks per wave (4 MFMAs per A
// fragment instead of 2; 64 points per wave): the upper bound of what more reuse of an A fragment would buy.
template <int AG, int NBK>
__global__ __launch_bounds__(512) void k_m2(int chunks, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * 16384 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * (i & 255);
    h8 B[8][NBK];
    for (int k = 0; k < 8; ++k) for (int nb = 0; nb < NBK; ++nb) for (int e = 0; e < 8; ++e) B[k][nb][e] = (_Float16)(0.01f * (k + nb + e + lane));
    __syncthreads();
    constexpr int PF = 3, QN = 4;
    h8 aq[QN];
    f4 acc[2][NBK];
    for (int m = 0; m < 2; ++m) for (int nb = 0; nb < NBK; ++nb) acc[m][nb] = (f4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        const unsigned slot = (unsigned)(size_t)(smem + (c % 3) * 16384 + lane * 16);
#define LD(dst, off) do { if (AG) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(slot), "n"(off)); \
                          else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(slot), "n"(off)); } while (0)
#pragma unroll
        for (int t = 0; t < PF; ++t) LD(aq[t % QN], t * 1024);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t + PF < 16) LD(aq[(t + PF) % QN], (t + PF) * 1024);
            // counted wait: everything but the PF (or fewer) reads just issued
            if (t + PF < 16) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else if (t + PF == 16) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else if (t + PF == 17) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int ks = t / 2, mbl = t % 2;
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
                if (AG) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[mbl][nb]) : "a"(aq[t % QN]), "v"(B[ks][nb]));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[mbl][nb]) : "v"(aq[t % QN]), "v"(B[ks][nb]));
            }
        }
#undef LD
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int nb = 0; nb < NBK; ++nb) s += acc[m][nb][0];
    if (s == 12345.678f) sink[0] = s;
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern) {
    const int chunks = 2000;
    unsigned long long* cyc; float* sink;
    (void)hipMalloc(&cyc, 64); (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
    printf("%-52s", name);
    for (int threads : {256, 512}) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 3 * 16384, 0, 10, cyc, sink);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 3 * 16384, 0, chunks, cyc, sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf(" | %d w/SIMD: wave0 %6.1f cyc/chunk, wall %.3f ms (%.0f ns per chunk per SIMD)", threads / 256, (double)c[0] / chunks, ms,
               ms * 1e6 / chunks / (threads / 256));
    }
    printf("   [MFMA pipe: 512 cyc per chunk per wave]\n");
}
int main() {
    run("no LDS, one B quad, 16x16x32", k_m<0, 0, 3, 0>);
    run("no LDS, 16 B quads, 16x16x32", k_m<0, 1, 3, 0>);
    run("LDS tiles PF 3, one B quad", k_m<1, 0, 3, 0>);
    run("LDS tiles PF 3, 16 B quads", k_m<1, 1, 3, 0>);
    run("LDS tiles PF 6, 16 B quads", k_m<1, 1, 6, 0>);
    run("no LDS, 8 B quads, 32x32x16", k_m<0, 1, 3, 1>);
    run("LDS tiles PF 3, 8 B quads, 32x32x16", k_m<1, 1, 3, 1>);
    run("LDS tiles PF 6, 8 B quads, 32x32x16", k_m<1, 1, 6, 1>);
    printf("-- round 5: asm stream (counted waits), 16 tiles per chunk; MFMA pipe per chunk per wave: 512 cyc (NB 2), 1024 (NB 4)\n");
    run("asm, A frags in VGPRs, NB 2", k_m2<0, 2>);
    run("asm, A frags in AGPRs (ds_read_b128 a[..]), NB 2", k_m2<1, 2>);
    run("asm, A frags in VGPRs, NB 4 (64 points per wave)", k_m2<0, 4>);
    run("asm, A frags in AGPRs, NB 4", k_m2<1, 4>);
    return 0;
}
