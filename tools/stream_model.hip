// Machine model of the fused-MLP stream STRUCTURES (standalone; no product code, no real data):
//   hipcc --offload-arch=gfx950 -O3 -o stream_model tools/stream_model.hip && ./stream_model
// One workgroup per CU runs `layers` 256x256 layers for a tile of points with the real instruction mix of the shading
// pair -- MFMAs, A tiles from an LDS ring, the softplus+sigmoid activation program (forward sweep) or the sigmoid
// multiplication (reverse sweep) in lock step, optionally (FULL) the weight DMA, the per-chunk barrier and the sigmoid
// stores / loads -- in four shapes:
//   pp2  : 8 waves, 32 points each, MFMA 16x16x32, phase-separated M(c) | V(c), the two waves of a SIMD in anti-phase
//          (= mlp_core.hpp's run_layer_pp today; calibrates the model against the product kernels)
//   pp32 : the same with MFMA 32x32x16 (one 32-point column block per wave, 64-row chunks)
//   il4  : 4 waves (one per SIMD, 512 registers), 64 points each as 4 column blocks of 16x16x32; the activation of
//          chunk c-1 is issued between the MFMAs of chunk c
//   il32 : 4 waves, 64 points each as 2 column blocks of 32x32x16, 64-row chunks, same interleave
// Activation mixes: fwd / rev with half-precision sigmoids through HBM (what the product does), fwd8 / rev8 with 8-bit
// sigmoids (unorm8: round(255 s) read off the mantissa of 1024 + 255 s, bytes gathered with v_perm_b32) either through
// HBM at half the bytes or ON CHIP in accumulator registers (`oc`: the fused forward+reverse tile, 32 points per wave).
// Reported: shader cycles per layer (wave 0), wall time, MFMA rate in TFLOP/s and as a fraction of 2.5 PFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define SB __builtin_amdgcn_sched_barrier(0)
#define DEV __device__ __forceinline__

template <int BIG_, int RB_, int CB_, int KS_, bool IL_>
struct Geom {
    static constexpr int BIG = BIG_, RB = RB_, CB = CB_, KS = KS_;
    static constexpr bool IL = IL_;
    static constexpr int WAVES = IL ? 4 : 8;
    static constexpr int KPER = BIG ? 16 : 32, RPB = BIG ? 32 : 16, PPB = BIG ? 32 : 16;   // K per step, rows / points per block
    static constexpr int GPA = BIG ? 8 : 2;                 // packed row pairs per accumulator
    static constexpr int ROWS = RB * RPB, PTS = CB * PPB;   // chunk
    static constexpr int NM = KS * RB * CB, NT = KS * RB;   // MFMAs, A tiles per chunk
    static constexpr int NG = RB * CB * GPA;                // row-pair groups per chunk = dwords of next-layer operand per lane
    static constexpr int CPL = 256 / ROWS;                  // chunks per layer
    static constexpr int KPC = ROWS / KPER;                 // K steps of the next layer one chunk produces
    static constexpr int CHB = NT * 1024;                   // chunk bytes in the ring
    static constexpr int PF = 3, QN = 4;
    static_assert(KS * KPER == 256 && NG == KPC * CB * 4, "geometry");
    typedef typename std::conditional<BIG != 0, f16v, f4>::type acc_t;
};

DEV unsigned bits(h2 v) { return __builtin_bit_cast(unsigned, v); }
DEV h2 to_h2(float a, float b) { return __builtin_convertvector((f2){a, b}, h2); }

template <int NG>
struct VR { unsigned z[NG], u[NG], t[NG], r[NG], d[NG]; };
struct KC { unsigned c1, c2, c3, c255, c1024, selp, sel0, sel1, c64; };

// instruction I of the lock-step activation program over NG row-pair groups: stage I / NG, group I % NG
// MIX 0 = forward sweep (softplus + stored sigmoid: 10 stages), 1 = reverse sweep (2 stages),
//     2 = forward with 8-bit sigmoids (+ scale, + byte gather per two groups, + accvgpr_write per two groups when on chip),
//     3 = reverse with 8-bit sigmoids (accvgpr_read per two groups when on chip, byte spread, - 1024, multiply)
template <int MIX, bool ONCHIP, int NG, int I>
DEV void emit(VR<NG>& v, const KC& k, const float (&af)[2 * NG], unsigned (&bn)[NG], unsigned (&sg)[NG], unsigned (&sa)[NG / 2]) {
    constexpr int st = I / NG, q = I % NG;
    if constexpr (st == 0) {
        v.z[q] = bits(to_h2(af[2 * q], af[2 * q + 1]));
    } else if constexpr (MIX == 1) {
        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(bn[q]) : "v"(v.z[q]), "v"(sg[q]));
    } else if constexpr (MIX == 3) {
        if constexpr (st == 1) {
            if constexpr (ONCHIP && q % 2 == 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sg[q]) : "a"(sa[q / 2]));
        } else if constexpr (st == 2) {   // bytes (2 (q%2), 2 (q%2) + 1) of the pair's dword -> halves 0x64bb = 1024 + b
            if constexpr (q % 2 == 0) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v.t[q]) : "v"(sg[q]), "v"(k.c64), "v"(k.sel0));
            else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v.t[q]) : "v"(sg[q - 1]), "v"(k.c64), "v"(k.sel1));
        } else if constexpr (st == 3) {
            asm volatile("v_pk_add_f16 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(v.t[q]) : "v"(k.c1024));
        } else {
            static_assert(st == 4, "stage");
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(bn[q]) : "v"(v.z[q]), "v"(v.t[q]));
        }
    } else if constexpr (st == 1) {
        asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(v.u[q]) : "v"(v.z[q]));
    } else if constexpr (st == 2) {
        asm volatile("v_exp_f16_sdwa %0, -|%1| dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(v.u[q]) : "v"(v.z[q]));
    } else if constexpr (st == 3) {
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(v.t[q]) : "v"(v.u[q]), "v"(k.c3), "v"(k.c2));
    } else if constexpr (st == 4) {
        asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(v.t[q]) : "v"(v.u[q]), "v"(k.c1));
    } else if constexpr (st == 5) {
        asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(v.r[q]) : "v"(v.z[q]));
    } else if constexpr (st == 6) {
        asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(bn[q]) : "v"(v.t[q]), "v"(v.u[q]), "v"(v.r[q]));
    } else if constexpr (st == 7) {
        asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v.d[q]) : "v"(v.z[q]), "v"(bn[q]));
    } else if constexpr (st == 8) {
        asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=&v"(sg[q]) : "v"(v.d[q]));
    } else if constexpr (st == 9) {
        asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(sg[q]) : "v"(v.d[q]));
    } else if constexpr (st == 10) {   // 1024 + 255 s: the mantissa's low byte is round(255 s)
        static_assert(MIX == 2, "stage");
        asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(sg[q]) : "v"(k.c255), "v"(k.c1024));
    } else if constexpr (st == 11) {   // the four low bytes of two groups into one dword
        if constexpr (q % 2 == 0) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(sg[q]) : "v"(sg[q + 1]), "0"(sg[q]), "v"(k.selp));
    } else {
        static_assert(st == 12 && ONCHIP, "stage");
        if constexpr (q % 2 == 0) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(sa[q / 2]) : "v"(sg[q]));
    }
}
template <int MIX, bool ONCHIP> constexpr int n_stages() { return MIX == 0 ? 10 : MIX == 1 ? 2 : MIX == 2 ? (ONCHIP ? 13 : 12) : 5; }
template <int MIX, bool ONCHIP, int NG, int I0, int I1>
DEV void emit_range(VR<NG>& v, const KC& k, const float (&af)[2 * NG], unsigned (&bn)[NG], unsigned (&sg)[NG], unsigned (&sa)[NG / 2]) {
    if constexpr (I0 < I1) {
        emit<MIX, ONCHIP, NG, I0>(v, k, af, bn, sg, sa);
        emit_range<MIX, ONCHIP, NG, I0 + 1, I1>(v, k, af, bn, sg, sa);
    }
}

DEV void dma_piece(const char* gsrc_uniform, unsigned lane_off, unsigned lds_base_uniform) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base_uniform), "v"(lane_off), "s"(gsrc_uniform)
                 : "memory");
}
DEV const char* uniform_ptr(const char* p) {
    const size_t v = (size_t)p;
    return (const char*)(((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

template <class G>
struct State {
    h8 Bcur[G::KS][G::CB];
    unsigned Bn[G::CPL][G::NG];
    typename G::acc_t acc[2][G::RB][G::CB];
    h8 aq[G::QN];
    u4 dreg[4];                  // weight pieces in flight through registers (DMA style 16)
    unsigned sa[2][G::NG / 2];   // on-chip 8-bit sigmoids (accumulator registers)
    unsigned sg[2][G::NG];   // sigmoid fragments: produced (forward) / loaded (reverse); double-buffered for the interleave
    VR<G::NG> v;
    KC k;
    const char* ring;
    const char* wts;
    char* sigp;      // this wave's sigmoid block of the current layer
    int ring_pos, lane, wave;
    unsigned long long chunk_no;
};

template <class G>
DEV typename G::acc_t mfma(h8 a, h8 b, typename G::acc_t c) {
    if constexpr (G::BIG) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
template <class G>
DEV void flatten(const typename G::acc_t (&acc)[G::RB][G::CB], float (&af)[2 * G::NG]) {
#pragma unroll
    for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
        for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
            for (int e = 0; e < 2 * G::GPA; ++e) af[(rb * G::CB + cb) * 2 * G::GPA + e] = acc[rb][cb][e];
}
template <class G>
DEV h8 lds_tile(const char* slot, int t, int lane) { return *(const h8*)(slot + t * 1024 + lane * 16); }
// operand registers of K steps [c KPC, (c+1) KPC) from chunk c's outputs
template <class G, int C>
DEV void adopt(State<G>& s) {
#pragma unroll
    for (int i = 0; i < G::KPC; ++i)
#pragma unroll
        for (int cb = 0; cb < G::CB; ++cb) {
            u4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = s.Bn[C][(i * G::CB + cb) * 4 + e];
            s.Bcur[C * G::KPC + i][cb] = __builtin_bit_cast(h8, w);
        }
}
// one M0 write, the wave's four pieces contiguous, addressed through the instruction offset
template <class G>
DEV void dma_chunk_imm(State<G>& s) {
    const unsigned slot_free = (s.ring_pos + 2) % 3;
    const int wv = __builtin_amdgcn_readfirstlane(s.wave) % 4;
    const char* src = uniform_ptr(s.wts) + (size_t)((s.chunk_no + 2) % 48) * G::CHB + wv * 4096;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)s.ring) + slot_free * G::CHB + wv * 4096;
    const unsigned lo = s.lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(dst), "v"(lo), "s"(src) : "memory");
}
template <class G, int FULL>
DEV void dma_chunk_pieces(State<G>& s, int first, int n) {   // pieces [first, first+n) of this wave's share of the next chunk
    if constexpr ((FULL & 2) != 0) {
        const unsigned slot_free = (s.ring_pos + 2) % 3;   // consumed in the previous chunk
        const char* src = uniform_ptr(s.wts) + (size_t)((s.chunk_no + 2) % 48) * G::CHB;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)s.ring) + slot_free * G::CHB;
        const int wv = __builtin_amdgcn_readfirstlane(s.wave) % 4;
        for (int i = first; i < first + n; ++i) {
            const int piece = wv + 4 * i;
            dma_piece(src + piece * 1024, s.lane * 16, dst + piece * 1024);
        }
    }
}

// FULL bits: 1 chunk barrier, 2 weight DMA, 4 sigmoids through HBM (8-bit mixes without it: on chip), 8 counted waits
// (the forward sweep's stores are not waited for; the reverse sweep's loads run one chunk ahead)
template <int MIX, int FULL> constexpr bool onchip() { return MIX >= 2 && !(FULL & 4); }
template <class G, int MIX> constexpr int sig_vec() { return MIX >= 2 ? G::NG / 8 : G::NG / 4; }   // 16-byte vectors per lane per chunk
template <int N> DEV void wait_vm() {   // s_waitcnt vmcnt(N), N < 64 (gfx9 encoding: bits 3:0 and 15:14)
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
template <class G, int MIX, int FULL>
DEV void sig_load(State<G>& s, int c, unsigned (&sg)[G::NG]) {
    if constexpr ((FULL & 4) && (MIX == 1 || MIX == 3)) {
        constexpr int NV = sig_vec<G, MIX>(), STEP = MIX == 3 ? 2 : 1;   // 8-bit: one dword per two groups, kept in the even slots
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const u4 w = *(const u4*)(s.sigp + (c * NV + i) * 1024 + s.lane * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) sg[(4 * i + e) * STEP] = w[e];
        }
    }
}
template <class G, int MIX, int FULL>
DEV void sig_store(State<G>& s, int c, unsigned (&sg)[G::NG]) {
    if constexpr ((FULL & 4) && (MIX == 0 || MIX == 2)) {
        constexpr int NV = sig_vec<G, MIX>(), STEP = MIX == 2 ? 2 : 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            u4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = sg[(4 * i + e) * STEP];
            *(u4*)(s.sigp + (c * NV + i) * 1024 + s.lane * 16) = w;
        }
    }
}

// ---------------------------------------------------------------------------------------- interleaved (one wave / SIMD)
template <class G, int MIX, int FULL, int C, int M>
DEV void il_step(State<G>& s, const float (&af)[2 * G::NG], const char* slot, const char* slot_next) {
    if constexpr (M < G::NM) {
        constexpr int ks = M / (G::RB * G::CB), rb = (M / G::CB) % G::RB, cb = M % G::CB, t = M / G::CB;
        constexpr int buf = C % 2, prev = (C + G::CPL - 1) % G::CPL;
        // V(prev) must be complete before the K steps it feeds: in a layer's first chunk those are the last KPC K steps
        constexpr int SPAN = C == 0 ? G::NM - G::KPC * G::RB * G::CB : G::NM;
        constexpr int NV = n_stages<MIX, onchip<MIX, FULL>()>() * G::NG;
        if constexpr (cb == 0) {
            if constexpr (t + G::PF < G::NT) s.aq[(t + G::PF) % G::QN] = lds_tile<G>(slot, t + G::PF, s.lane);
            else s.aq[(t + G::PF) % G::QN] = lds_tile<G>(slot_next, t + G::PF - G::NT, s.lane);
        }
        if constexpr (C == 0 && M == SPAN) adopt<G, G::CPL - 1>(s);
        SB;
        const h8 a = s.aq[t % G::QN];
        if constexpr (ks == 0) s.acc[buf][rb][cb] = mfma<G>(a, s.Bcur[ks][cb], (typename G::acc_t)(0.0f));
        else s.acc[buf][rb][cb] = mfma<G>(a, s.Bcur[ks][cb], s.acc[buf][rb][cb]);
        constexpr int DSTEP = G::NM / (G::NT / 4) / 2 > 0 ? G::NM / (G::NT / 4) / 2 : 1;   // this wave's NT/4 pieces over the chunk's first half
        if constexpr ((FULL & 2) && M % DSTEP == 1 && M / DSTEP < G::NT / 4) dma_chunk_pieces<G, FULL>(s, M / DSTEP, 1);
        if constexpr (M < SPAN) {
            constexpr int i0 = (int)((long long)M * NV / SPAN), i1 = (int)((long long)(M + 1) * NV / SPAN);
            emit_range<MIX, onchip<MIX, FULL>(), G::NG, i0, i1>(s.v, s.k, af, s.Bn[prev], s.sg[1 - buf], s.sa[1 - buf]);
        }
        SB;
        il_step<G, MIX, FULL, C, M + 1>(s, af, slot, slot_next);
    }
}
template <class G, int MIX, int FULL, int C>
DEV void il_chunk(State<G>& s) {
    constexpr int buf = C % 2, prev = (C + G::CPL - 1) % G::CPL;
    const char* slot = s.ring + s.ring_pos * G::CHB;
    const char* slot_next = s.ring + ((s.ring_pos + 1) % 3) * G::CHB;
    float af[2 * G::NG];
    flatten<G>(s.acc[1 - buf], af);
    sig_load<G, MIX, FULL>(s, C, s.sg[buf]);   // this chunk's stored sigmoids, used by V(C) inside chunk C + 1
    il_step<G, MIX, FULL, C, 0>(s, af, slot, slot_next);
    sig_store<G, MIX, FULL>(s, prev, s.sg[1 - buf]);   // the sigmoids V(prev) just produced
    if constexpr ((FULL & 6) != 0) {
        // the DMA pieces were issued in the chunk's first half; the stores behind them need no wait
        if constexpr ((FULL & 8) && (MIX == 0 || MIX == 2) && (FULL & 4)) wait_vm<sig_vec<G, MIX>()>();
        else wait_vm<0>();
    }
    if constexpr ((FULL & 1) != 0) __syncthreads();
    s.ring_pos = (s.ring_pos + 1) % 3;
    ++s.chunk_no;
}

// ---------------------------------------------------------------------------------------- phase-separated (two waves / SIMD)
template <class G, int MIX, int FULL, int C>
DEV void pp_chunk(State<G>& s) {
    const char* slot = s.ring + s.ring_pos * G::CHB;
    const char* slot_next = s.ring + ((s.ring_pos + 1) % 3) * G::CHB;
    const bool late = __builtin_amdgcn_readfirstlane(s.wave) >= 4;
    constexpr bool CW = (FULL & 8) != 0, REV = MIX == 1 || MIX == 3, HBM = (FULL & 4) != 0;
    constexpr int sb = (CW && REV) ? C % 2 : 0;
    if constexpr (CW && REV) sig_load<G, MIX, FULL>(s, (C + 1) % G::CPL, s.sg[1 - sb]);   // one chunk ahead
    else sig_load<G, MIX, FULL>(s, C, s.sg[0]);
#pragma unroll
    for (int t = 0; t < G::NT; ++t) {
        const int ks = t / G::RB, rb = t % G::RB;
        const h8 a = s.aq[t % G::QN];
        if (t + G::PF < G::NT) s.aq[(t + G::PF) % G::QN] = lds_tile<G>(slot, t + G::PF, s.lane);
        SB;
#pragma unroll
        for (int cb = 0; cb < G::CB; ++cb) {
            if (ks == 0) s.acc[0][rb][cb] = mfma<G>(a, s.Bcur[ks][cb], (typename G::acc_t)(0.0f));
            else s.acc[0][rb][cb] = mfma<G>(a, s.Bcur[ks][cb], s.acc[0][rb][cb]);
        }
        SB;
    }
    // in flight, oldest first: [loads of this chunk (CW: issued a chunk ago)] [late waves: DMA pieces] [CW: loads of the next chunk]
    if constexpr (HBM && REV) {
        if constexpr (CW) wait_vm<sig_vec<G, MIX>()>();
        else wait_vm<0>();
    }
    if (late) {
        if constexpr ((FULL & 2) != 0 && !(FULL & 16) && !(HBM && REV)) {
            if constexpr (CW && HBM) wait_vm<sig_vec<G, MIX>()>();   // forward: the stores behind the DMA pieces stay in flight
            else wait_vm<0>();
        }
        if constexpr ((FULL & 1) != 0) __syncthreads();
        if constexpr ((FULL & 16) != 0) {   // through registers
            const char* src = s.wts + (size_t)((s.chunk_no + 2) % 48) * G::CHB + (s.wave % 4) * 4096 + s.lane * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) s.dreg[i] = *(const u4*)(src + i * 1024);
        } else if constexpr ((FULL & 32) != 0) {
            dma_chunk_imm<G>(s);
        } else {
            dma_chunk_pieces<G, FULL>(s, 0, G::NT / 4);
        }
    }
    SB;
    float af[2 * G::NG];
    flatten<G>(s.acc[0], af);
    emit_range<MIX, onchip<MIX, FULL>(), G::NG, 0, n_stages<MIX, onchip<MIX, FULL>()>() * G::NG>(s.v, s.k, af, s.Bn[C], s.sg[sb], s.sa[0]);
    sig_store<G, MIX, FULL>(s, C, s.sg[0]);
    if constexpr ((FULL & 16) != 0) {
        if (late) {
            char* dst = const_cast<char*>(s.ring) + ((s.ring_pos + 2) % 3) * G::CHB + (s.wave % 4) * 4096 + s.lane * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u4*)(dst + i * 1024) = s.dreg[i];
        }
    }
    SB;
    if (!late) {
        if constexpr ((FULL & 1) != 0) __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < G::PF; ++t) s.aq[t % G::QN] = lds_tile<G>(slot_next, t, s.lane);
    s.ring_pos = (s.ring_pos + 1) % 3;
    ++s.chunk_no;
}

template <class G, int MIX, int FULL, int C>
DEV void layer_chunks(State<G>& s) {
    if constexpr (C < G::CPL) {
        if constexpr (G::IL) il_chunk<G, MIX, FULL, C>(s);
        else pp_chunk<G, MIX, FULL, C>(s);
        layer_chunks<G, MIX, FULL, C + 1>(s);
    }
}
template <class G, int C, int N>
DEV void adopt_all(State<G>& s) {
    if constexpr (C < N) {
        adopt<G, C>(s);
        adopt_all<G, C + 1, N>(s);
    }
}

template <class G, int MIX, int FULL>
__global__ __launch_bounds__(G::WAVES * 64) void k_stream(int layers, const char* __restrict__ wts, char* __restrict__ sigbuf,
                                                           unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    State<G> s;
    s.lane = threadIdx.x & 63;
    s.wave = threadIdx.x >> 6;
    s.ring = smem;
    s.wts = wts;
    s.ring_pos = 0;
    s.chunk_no = 0;
    for (int i = threadIdx.x; i < 3 * G::CHB / 4; i += blockDim.x)
        ((unsigned*)smem)[i] = 0x20002000u + ((i * 2654435761u) >> 20 & 0x03ff03ffu) + ((i & 1) ? 0x80000000u : 0u);   // halves ~ +-0.01
    const h2 c1 = {(_Float16)1.42459527f, (_Float16)1.42459527f}, c2 = {(_Float16)-0.58921265f, (_Float16)-0.58921265f},
             c3 = {(_Float16)0.16538905f, (_Float16)0.16538905f};
    const h2 c255 = {(_Float16)255.0f, (_Float16)255.0f}, c1024 = {(_Float16)1024.0f, (_Float16)1024.0f};
    // v_perm_b32 D = bytes of {S0, S1} (S1 = bytes 0..3, S0 = bytes 4..7): gather = low bytes of both halves of both sources
    s.k = KC{bits(c1), bits(c2), bits(c3), bits(c255), bits(c1024), 0x06040200u, 0x05010500u, 0x05030502u, 0x64646464u};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
        for (int cb = 0; cb < G::CB; ++cb)
#pragma unroll
            for (int e = 0; e < 8; ++e) s.Bcur[ks][cb][e] = (_Float16)(0.25f * (((ks * 7 + cb * 3 + e + s.lane) % 13) - 6));
#pragma unroll
    for (int c = 0; c < G::CPL; ++c)
#pragma unroll
        for (int q = 0; q < G::NG; ++q) s.Bn[c][q] = 0x38003800u;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < G::CB; ++cb) s.acc[b][rb][cb] = (typename G::acc_t)(0.0f);
#pragma unroll
        for (int q = 0; q < G::NG; ++q) s.sg[b][q] = 0x38003800u;
#pragma unroll
        for (int q = 0; q < G::NG / 2; ++q) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(s.sa[b][q]) : "v"(0x80808080u + s.lane));
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < G::PF; ++t) s.aq[t % G::QN] = lds_tile<G>(smem, t, s.lane);
    const size_t layer_bytes = (size_t)G::CPL * G::NG * 256;                     // per wave per layer: 256 features x PTS points x 2 B
    const size_t wg_bytes = layer_bytes * G::WAVES;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < layers; ++l) {
        // streams through a 1 GiB window like the sigmoid buffer of a 2 M-point segment
        const size_t slot = ((size_t)(l % 32) * gridDim.x + blockIdx.x) % ((size_t)(1u << 30) / wg_bytes);
        s.sigp = sigbuf + slot * wg_bytes + s.wave * layer_bytes;
        layer_chunks<G, MIX, FULL, 0>(s);
        // interleaved: the last chunk's activation is still running; it is adopted inside the next layer's first chunk
        adopt_all<G, 0, G::IL ? G::CPL - 1 : G::CPL>(s);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < G::CB; ++cb) r += s.acc[b][rb][cb][0];
#pragma unroll
    for (int c = 0; c < G::CPL; ++c)
#pragma unroll
        for (int q = 0; q < G::NG; ++q) r += (float)s.Bn[c][q];
#pragma unroll
    for (int q = 0; q < G::NG; ++q) r += (float)(s.sg[0][q] + s.sg[1][q]);
#pragma unroll
    for (int q = 0; q < G::NG / 2; ++q) { unsigned t0, t1; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t0) : "a"(s.sa[0][q])); asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t1) : "a"(s.sa[1][q])); r += (float)(t0 + t1); }
    if (r == 12345.678f) sink[0] = r;
    if (blockIdx.x == 0 && s.lane == 0) cyc[s.wave] = t1 - t0;
}

static char* g_wts;
static char* g_sig;
static unsigned long long* g_cyc;
static float* g_sink;

template <class G, int MIX, int FULL>
void run(const char* name) {
    const int layers = 600;
    auto kern = k_stream<G, MIX, FULL>;
    const int lds = 3 * G::CHB;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(256), dim3(G::WAVES * 64), lds, 0, 20, g_wts, g_sig, g_cyc, g_sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%-40s launch failed: %s\n", name, hipGetErrorString(hipGetLastError())); return; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    unsigned long long c[8];
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(G::WAVES * 64), lds, 0, layers, g_wts, g_sig, g_cyc, g_sink);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    (void)hipMemcpy(c, g_cyc, 64, hipMemcpyDeviceToHost);
    const double flop = 2.0 * 256.0 * (G::WAVES * G::PTS) * 65536.0 * layers;
    const double tf = flop / (best * 1e-3) / 1e12;
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, (const void*)kern);
    printf("%-44s %5d pts/WG  %7.0f cyc/layer (wave0)  wall %7.3f ms  %7.1f TFLOP/s  frac %.3f  clock %.2f GHz  regs %d spill %d\n", name,
           G::WAVES * G::PTS, (double)c[0] / layers, best, tf, tf / 2500.0, c[0] / (best * 1e6), fa.numRegs, (int)fa.localSizeBytes);
    fflush(stdout);
}

int main(int argc, char** argv) {
    (void)hipMalloc(&g_wts, 48 * 32768);
    (void)hipMemset(g_wts, 0x11, 48 * 32768);
    (void)hipMalloc(&g_sig, (size_t)1 << 30);
    (void)hipMemset(g_sig, 0x38, (size_t)1 << 30);
    (void)hipMalloc(&g_cyc, 64);
    (void)hipMalloc(&g_sink, 4);
    typedef Geom<0, 2, 2, 8, false> PP2;
    typedef Geom<1, 2, 1, 16, false> PP32;
    typedef Geom<0, 2, 4, 8, true> IL4;
    typedef Geom<1, 2, 2, 16, true> IL32;
    const char* only = argc > 1 ? argv[1] : "";
#define RUN(G, MIX, FULL, name) if (!only[0] || strstr(name, only)) run<G, MIX, FULL>(name)
    RUN(PP2, 0, 1, "pp2  fwd   +barrier");
    RUN(PP2, 0, 3, "pp2  fwd   +barrier+dma (lds-dma, M0 per piece)");
    RUN(PP2, 0, 3 + 32, "pp2  fwd   +barrier+dma (lds-dma, one M0, offsets)");
    RUN(PP2, 0, 3 + 16, "pp2  fwd   +barrier+dma (through registers)");
    RUN(PP2, 1, 1, "pp2  rev   +barrier");
    RUN(PP2, 1, 3, "pp2  rev   +barrier+dma (lds-dma, M0 per piece)");
    RUN(PP2, 1, 3 + 32, "pp2  rev   +barrier+dma (lds-dma, one M0, offsets)");
    RUN(PP2, 1, 3 + 16, "pp2  rev   +barrier+dma (through registers)");
    RUN(PP2, 0, 7, "pp2  fwd   full (product)");
    RUN(PP2, 0, 7 + 32, "pp2  fwd   full, one M0");
    RUN(PP2, 0, 7 + 16, "pp2  fwd   full, through registers");
    RUN(PP2, 1, 15, "pp2  rev   full, loads one chunk ahead");
    RUN(PP2, 1, 15 + 32, "pp2  rev   full, ahead, one M0");
    RUN(PP2, 3, 15, "pp2  rev8  full, ahead");
    RUN(PP2, 3, 15 + 32, "pp2  rev8  full, ahead, one M0");
    RUN(PP2, 2, 7 + 32, "pp2  fwd8  full, one M0");
    return 0;
}
