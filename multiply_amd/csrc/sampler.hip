// VolSDF error-bound sampler (reference code/lib/model/ray_sampler.py:66-230), split at the SDF queries.
// One wave (64 lanes) owns one ray; its sorted sample list (<= n_eval * max_iters entries) lives in LDS and the
// cumulative sums of Algorithm 1 are wave-level scans.  Entry points: include/multiply_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/multiply_hip.h"
#include "common.hpp"

using namespace mp;

namespace {

constexpr int WAVES = 4;

// orders this wave's LDS traffic: writes by any lane before, reads by any lane after
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The sampler's exponentials and densities in the hardware's own precision: v_exp_f32 of x * log2(e) (relative error about
// |x| * 6e-8, i.e. < 6e-6 for the arguments that matter) and a multiplication by 1 / beta instead of three IEEE
// divisions.  The sampler decides where to put samples from f16 SDF queries (its depths are compared with the oracle at
// 3e-4): libm's last-ulp exp / expm1 and correctly rounded divisions were most of the instructions of these VALU-bound
// kernels.  The compositing kernels, whose outputs are compared at 1e-6, keep the exact forms of common.hpp.
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
// LaplaceDensity (density.py:20-29): (1/beta) (0.5 + 0.5 sign(s) expm1(-|s|/beta)) = (1/beta) * {0.5 e, 1 - 0.5 e, 0.5}
__device__ __forceinline__ float density(float sdf, float inv_beta) {
    const float e = fexp(-fabsf(sdf) * inv_beta);
    // a NaN sdf (a broken network evaluation) stays NaN, as in the reference's sign() * expm1() form: it must surface in the
    // error bound (nan_max), not turn into the density of the surface
    return sdf != sdf ? sdf : inv_beta * (sdf > 0.0f ? 0.5f * e : (sdf < 0.0f ? 1.0f - 0.5f * e : 0.5f));
}

struct Arr {
    float *Z, *S, *D, *A, *B;
};

__device__ __forceinline__ Arr wave_arrays(char* smem, int zm) {
    float* base = (float*)smem + (size_t)(threadIdx.x >> 6) * 5 * zm;
    return Arr{base, base + zm, base + 2 * zm, base + 3 * zm, base + 4 * zm};
}

// torch.linspace(start, end, steps)[i] in fp32 (symmetric evaluation used by ATen's CPU and CUDA kernels)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
    if (steps == 1) return start;
    const float step = (end - start) / (float)(steps - 1);
    return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// number of entries of sorted a[0..n) that are < v (strict) or <= v
__device__ __forceinline__ int count_less(const float* a, int n, float v, bool or_equal) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const bool go = or_equal ? (a[mid] <= v) : (a[mid] < v);
        if (go) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// d* of Theorem 1 (ray_sampler.py:98-110) for interval i, from Z/S
__device__ __forceinline__ float d_star(const float* Z, const float* S, int i) {
    const float a = Z[i + 1] - Z[i], s0 = S[i], s1 = S[i + 1];
    const float b = fabsf(s0), c = fabsf(s1);
    const bool first = a * a + b * b <= c * c, second = a * a + c * c <= b * b;
    float d = 0.0f;
    if (first) d = b;
    if (second) d = c;
    if (!first && !second && (b + c - a > 0.0f)) {
        const float s = (a + b + c) / 2.0f;
        const float area = s * (s - a) * (s - b) * (s - c);
        d = (2.0f * sqrtf(area)) / a;
    }
    const float sg0 = s0 > 0.f ? 1.f : (s0 < 0.f ? -1.f : 0.f), sg1 = s1 > 0.f ? 1.f : (s1 < 0.f ? -1.f : 0.f);
    return (sg1 * sg0 == 1.0f) ? d : 0.0f * d;  // (sign product == 1) * d_star; keeps NaN like the reference
}

__device__ __forceinline__ float nan_max(float m, float v) { return (m != m || v != v) ? NAN : fmaxf(m, v); }

// get_error_bound (ray_sampler.py:222-230): max over the n-1 intervals; wave-uniform result.
// Lane l owns the intervals [l*ch, l*ch + cnt); their (dist, sdf, d*) do not change over the beta line search, so the caller
// loads them into registers once (ChunkRegs) and every evaluation works from there: per interval one density and one
// exponential, each computed once and used by both the prefix sums and the running bound.
constexpr int MAXCH = 10;   // 640 samples / 64 lanes

struct ChunkRegs {
    float dist[MAXCH], s[MAXCH], d[MAXCH];
    int cnt;
};

__device__ __forceinline__ void load_chunk(const Arr& r, int n, ChunkRegs& c) {
    const int lane = threadIdx.x & 63, ni = n - 1;
    const int ch = (ni + 63) / 64, i0 = min(lane * ch, ni), i1 = min(i0 + ch, ni);
    c.cnt = i1 - i0;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const bool on = j < c.cnt;
        const int i = on ? i0 + j : 0;
        c.dist[j] = on ? r.Z[i + 1] - r.Z[i] : 0.f;
        c.s[j] = on ? r.S[i] : 0.f;
        c.d[j] = on ? r.D[i] : 0.f;
    }
}

__device__ float error_bound(const ChunkRegs& c, float beta) {
    const float ib = 1.0f / beta, inv4b2 = 0.25f * ib * ib;
    float fe[MAXCH], ee[MAXCH];
    float sfe = 0.f, serr = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)
        if (j < c.cnt) {
            fe[j] = c.dist[j] * density(c.s[j], ib);
            ee[j] = fexp(-c.d[j] * ib) * (c.dist[j] * c.dist[j]) * inv4b2;
            sfe += fe[j];
            serr += ee[j];
        }
    float tot;
    float integ = wave_excl_scan(sfe, tot);
    float errint = wave_excl_scan(serr, tot);
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j)
        if (j < c.cnt) {
            errint += ee[j];
            const float bound = (fminf(fexp(errint), 1.0e6f) - 1.0f) * fexp(-integ);
            m = nan_max(m, bound);
            integ += fe[j];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o));
    return m;
}

// the same from LDS, for sample counts beyond MAXCH * 64
__device__ float error_bound_lds(const Arr& r, int n, float beta) {
    const int lane = threadIdx.x & 63, ni = n - 1;
    const int ch = (ni + 63) / 64, i0 = min(lane * ch, ni), i1 = min(i0 + ch, ni);
    const float ib = 1.0f / beta, inv4b2 = 0.25f * ib * ib;
    float sfe = 0.f, serr = 0.f;
    for (int i = i0; i < i1; ++i) {
        const float dist = r.Z[i + 1] - r.Z[i];
        sfe += dist * density(r.S[i], ib);
        serr += fexp(-r.D[i] * ib) * (dist * dist) * inv4b2;
    }
    float tot;
    float integ = wave_excl_scan(sfe, tot);
    float errint = wave_excl_scan(serr, tot);
    float m = -INFINITY;
    for (int i = i0; i < i1; ++i) {
        const float dist = r.Z[i + 1] - r.Z[i];
        errint += fexp(-r.D[i] * ib) * (dist * dist) * inv4b2;
        const float bound = (fminf(fexp(errint), 1.0e6f) - 1.0f) * fexp(-integ);
        m = nan_max(m, bound);
        integ += dist * density(r.S[i], ib);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = nan_max(m, __shfl_xor(m, o));
    return m;
}

// ------------------------------------------------------------------------------------------------ init
__global__ __launch_bounds__(256) void k_sampler_init(MpSamplerCfg cfg, MpSamplerState st, const float* __restrict__ far,
                                                      const int* __restrict__ hit_index,
                                                      const int* __restrict__ hit_count, int max_rays, int n_groups,
                                                      const float* __restrict__ t_rand) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int n_rays = min(*hit_count, max_rays);
    const int NE = cfg.n_samples_eval;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < (cfg.max_total_iters + 1) * n_groups; i += 256) st.group_flag[i] = 0;
        for (int i = threadIdx.x; i <= cfg.max_total_iters; i += 256) st.any_active[i] = i == 0 ? 1 : 0;
    }
    float* zl = (float*)smem + (threadIdx.x >> 6) * NE;   // wave-private
    for (int k = blockIdx.x * WAVES + (threadIdx.x >> 6); k < n_rays; k += gridDim.x * WAVES) {
        const float fr = far[hit_index[k]], nr = cfg.near_;
        for (int i = lane; i < NE; i += 64) {
            const float t = linspace_at(0.0f, 1.0f, NE, i);
            zl[i] = nr * (1.0f - t) + fr * t;  // ray_sampler.py:29-30
        }
        wave_sync();
        float zj[4];  // NE <= 256
        for (int q = 0, i = lane; i < NE; i += 64, ++q) {
            float z = zl[i];
            if (t_rand) {  // stratified jitter, ray_sampler.py:32-40
                const float lower = i == 0 ? zl[0] : 0.5f * (zl[i] + zl[i - 1]);
                const float upper = i == NE - 1 ? zl[NE - 1] : 0.5f * (zl[i + 1] + zl[i]);
                z = lower + (upper - lower) * t_rand[(size_t)k * NE + i];
            }
            zj[q] = z;
            st.znew[(size_t)k * NE + i] = z;
        }
        wave_sync();
        for (int q = 0, i = lane; i < NE; i += 64, ++q) zl[i] = zj[q];
        wave_sync();
        // beta from the upper bound (ray_sampler.py:74-76) on the (jittered) depths
        float ss = 0.f;
        for (int i = lane; i < NE - 1; i += 64) { const float d = zl[i + 1] - zl[i]; ss += d * d; }
        ss = wsum(ss);
        if (lane == 0) {
            st.beta[k] = sqrtf((1.0f / (4.0f * logf(cfg.eps + 1.0f))) * ss);
            st.nz[k] = 0;
            st.ray_active[k] = 1;
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------ bound
__global__ __launch_bounds__(256) void k_sampler_bound(MpSamplerCfg cfg, MpSamplerState st,
                                                       const float* __restrict__ beta0_p,
                                                       const int* __restrict__ hit_index,
                                                       const int* __restrict__ hit_count, int max_rays,
                                                       int group_size, int n_groups, int iter, int zm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Arr r = wave_arrays(smem, zm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_rays = min(*hit_count, max_rays);
    const int NE = cfg.n_samples_eval;
    const float beta0 = *beta0_p;
    const int k = blockIdx.x * WAVES + wave;
    if (k >= n_rays || !st.ray_active[k]) return;  // waves are independent: no block-level barrier below
    const int n_old = st.nz[k];
    const int n = n_old + NE;
    // merge the queried samples into the sorted list (ray_sampler.py:89-94, 191: sort of cat == merge)
    for (int i = lane; i < n_old; i += 64) { r.A[i] = st.zs[(size_t)k * zm + i]; r.B[i] = st.sdfs[(size_t)k * zm + i]; }
    for (int i = lane; i < NE; i += 64) {
        r.A[n_old + i] = st.znew[(size_t)k * NE + i];
        r.B[n_old + i] = st.sdfnew[(size_t)k * NE + i];
    }
    wave_sync();
    for (int i = lane; i < n; i += 64) {
        const float zi = r.A[i];
        const int rank = i < n_old ? i + count_less(r.A + n_old, NE, zi, false)
                                   : (i - n_old) + count_less(r.A, n_old, zi, true);
        r.Z[rank] = zi;
        r.S[rank] = r.B[i];
    }
    wave_sync();
    for (int i = lane; i < n; i += 64) { st.zs[(size_t)k * zm + i] = r.Z[i]; st.sdfs[(size_t)k * zm + i] = r.S[i]; }
    for (int i = lane; i < n - 1; i += 64) r.D[i] = d_star(r.Z, r.S, i);
    wave_sync();
    // line search on beta (ray_sampler.py:113-122)
    float beta = st.beta[k];
    const bool in_regs = n - 1 <= MAXCH * 64;      // wave-uniform
    ChunkRegs cr;
    if (in_regs) load_chunk(r, n, cr);
    float curr = in_regs ? error_bound(cr, beta0) : error_bound_lds(r, n, beta0);
    if (curr <= cfg.eps) beta = beta0;
    float bmin = beta0, bmax = beta;
    for (int j = 0; j < cfg.beta_iters; ++j) {
        const float mid = (bmin + bmax) / 2.0f;
        curr = in_regs ? error_bound(cr, mid) : error_bound_lds(r, n, mid);
        if (curr <= cfg.eps) bmax = mid;
        if (curr > cfg.eps) bmin = mid;
    }
    beta = bmax;
    if (lane == 0) {
        st.beta[k] = beta;
        st.nz[k] = n;
        if (beta > beta0) atomicOr(&st.group_flag[iter * n_groups + hit_index[k] / group_size], 1);  // :137
    }
}

// ------------------------------------------------------------------------------------------------ resample
__global__ __launch_bounds__(256) void k_sampler_resample(MpSamplerCfg cfg, MpSamplerState st,
                                                          const float* __restrict__ beta0_p,
                                                          const float* __restrict__ far,
                                                          const int* __restrict__ hit_index,
                                                          const int* __restrict__ hit_count, int max_rays,
                                                          int group_size, int n_groups, int iter, int zm,
                                                          const float* __restrict__ u_final,
                                                          const int* __restrict__ extra_idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Arr r = wave_arrays(smem, zm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_rays = min(*hit_count, max_rays);
    const int k = blockIdx.x * WAVES + wave;
    if (k >= n_rays || !st.ray_active[k]) return;
    const int NE = cfg.n_samples_eval, NS = cfg.n_samples, NX = cfg.n_samples_extra;
    const int grp = hit_index[k] / group_size;
    const bool more = st.group_flag[iter * n_groups + grp] != 0 && (iter + 1 < cfg.max_total_iters);
    const int n = st.nz[k], ni = n - 1;
    const float beta = st.beta[k];
    for (int i = lane; i < n; i += 64) { r.Z[i] = st.zs[(size_t)k * zm + i]; r.S[i] = st.sdfs[(size_t)k * zm + i]; }
    wave_sync();
    // chunked scans over the n samples: free energy -> transmittance (ray_sampler.py:126-133)
    const int ch = (n + 63) / 64, i0 = min(lane * ch, n), i1 = min(i0 + ch, n);
    const float ib = 1.0f / beta, inv4b2 = 0.25f * ib * ib;
    // per-sample free energy and error term: computed once, kept in registers when the lane's chunk fits (ch <= MAXCH),
    // otherwise recomputed in the second sweep
    const bool in_regs = ch <= MAXCH;   // wave-uniform
    float fe_r[MAXCH], ee_r[MAXCH];
    float sfe = 0.f, serr = 0.f;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int i = i0 + j;
            fe_r[j] = 0.f; ee_r[j] = 0.f;
            if (i < i1) {
                const float dist = i < ni ? r.Z[i + 1] - r.Z[i] : 1e10f;
                fe_r[j] = dist * density(r.S[i], ib);
                sfe += fe_r[j];
                if (more && i < ni) {
                    ee_r[j] = fexp(-d_star(r.Z, r.S, i) * ib) * (dist * dist) * inv4b2;
                    serr += ee_r[j];
                }
            }
        }
    } else {
        for (int i = i0; i < i1; ++i) {
            const float dist = i < ni ? r.Z[i + 1] - r.Z[i] : 1e10f;
            sfe += dist * density(r.S[i], ib);
            if (more && i < ni) {
                r.D[i] = d_star(r.Z, r.S, i);
                serr += fexp(-r.D[i] * ib) * (dist * dist) * inv4b2;
            }
        }
    }
    float tot;
    float integ = wave_excl_scan(sfe, tot);
    float errint = wave_excl_scan(serr, tot);
    float psum = 0.f;
    auto emit = [&](int i, float fe, float ee) {
        const float trans = fexp(-integ);
        float pdf;
        if (more) {  // error-bound pdf (ray_sampler.py:142-149)
            if (i < ni) errint += ee;
            pdf = (fminf(fexp(errint), 1.0e6f) - 1.0f) * trans + cfg.add_tiny;
        } else {     // final pdf from the weights (ray_sampler.py:157-161)
            pdf = (1.0f - fexp(-fe)) * trans + 1e-5f;
        }
        if (i < ni) { r.A[i] = pdf; psum += pdf; }
        integ += fe;
    };
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < MAXCH; ++j)
            if (i0 + j < i1) emit(i0 + j, fe_r[j], ee_r[j]);
    } else {
        for (int i = i0; i < i1; ++i) {
            const float dist = i < ni ? r.Z[i + 1] - r.Z[i] : 1e10f;
            emit(i, dist * density(r.S[i], ib), (more && i < ni) ? fexp(-r.D[i] * ib) * (dist * dist) * inv4b2 : 0.f);
        }
    }
    const float total = wsum(psum);
    wave_sync();
    // cdf over the n bins edges: B[0] = 0, B[i+1] = cumsum(pdf/total)[i]
    const int chi = (ni + 63) / 64, j0 = min(lane * chi, ni), j1 = min(j0 + chi, ni);
    float loc = 0.f;
    for (int i = j0; i < j1; ++i) loc += r.A[i] / total;
    float run = wave_excl_scan(loc, tot);
    for (int i = j0; i < j1; ++i) { run += r.A[i] / total; r.B[i + 1] = run; }
    if (lane == 0) r.B[0] = 0.0f;
    wave_sync();
    // invert the cdf (ray_sampler.py:168-186)
    const int N = more ? NE : NS;
    float* outv = r.D;  // reuse: samples (and, for the final set, the extras) are collected here
    for (int j = lane; j < N; j += 64) {
        const float u = (more || !u_final) ? linspace_at(0.0f, 1.0f, N, j) : u_final[(size_t)k * NS + j];
        const int inds = count_less(r.B, n, u, true);  // searchsorted(right=True)
        const int below = max(inds - 1, 0), above = min(inds, n - 1);
        const float cb = r.B[below], ca = r.B[above];
        float denom = ca - cb;
        denom = denom < 1e-5f ? 1.0f : denom;
        const float t = (u - cb) / denom;
        outv[j] = r.Z[below] + t * (r.Z[above] - r.Z[below]);
    }
    wave_sync();
    if (more) {
        for (int j = lane; j < NE; j += 64) st.znew[(size_t)k * NE + j] = outv[j];
        if (lane == 0) st.any_active[iter + 1] = 1;
        return;
    }
    // final set: samples + near + far + N_extra of the current depths, sorted (ray_sampler.py:194-209)
    const int NF = NS + 2 + NX;
    if (lane == 0) { outv[NS] = cfg.near_; outv[NS + 1] = far[hit_index[k]]; }
    for (int j = lane; j < NX; j += 64) {
        // training: one randperm(n)[:NX] row per possible list length n = NE * k (ray_sampler.py:202)
        const int idx = extra_idx ? extra_idx[(n / NE - 1) * NX + j] : (int)linspace_at(0.0f, (float)(n - 1), NX, j);
        outv[NS + 2 + j] = r.Z[min(max(idx, 0), n - 1)];
    }
    wave_sync();
    // Stable sort of [samples | near | far | extras] (ray_sampler.py:208-209).  In eval mode the samples (ascending u) and the
    // extras (ascending indices into the sorted depths) are sorted runs already: an element's rank is its index in its own
    // run plus binary-search counts in the others (ties: the earlier run first, like the stable sort), 2 searches instead
    // of NF comparisons per element.  Random draws (training), or a run that rounding left out of order, take the rank sort.
    bool runs_sorted = !u_final && !extra_idx;
    if (runs_sorted) {
        bool ok = true;
        for (int j = lane; j < NF - 1; j += 64)
            if (j != NS - 1 && j != NS && j != NS + 1) ok = ok && outv[j] <= outv[j + 1];
        runs_sorted = __all(ok);
    }
    if (runs_sorted) {
        const float* Sr = outv;            // [NS]
        const float* Er = outv + NS + 2;   // [NX]
        const float vn = outv[NS], vf = outv[NS + 1];
        for (int j = lane; j < NF; j += 64) {
            const float v = outv[j];
            int rank;
            if (j < NS) rank = j + (vn < v ? 1 : 0) + (vf < v ? 1 : 0) + count_less(Er, NX, v, false);
            else if (j == NS) rank = count_less(Sr, NS, v, true) + (vf < v ? 1 : 0) + count_less(Er, NX, v, false);
            else if (j == NS + 1) rank = count_less(Sr, NS, v, true) + (vn <= v ? 1 : 0) + count_less(Er, NX, v, false);
            else rank = (j - NS - 2) + count_less(Sr, NS, v, true) + (vn <= v ? 1 : 0) + (vf <= v ? 1 : 0);
            st.zfinal[(size_t)k * NF + rank] = v;
        }
    } else {
        for (int j = lane; j < NF; j += 64) {  // rank sort (stable)
            const float v = outv[j];
            int rank = 0;
            for (int q = 0; q < NF; ++q) {
                const float w = outv[q];
                rank += (w < v || (w == v && q < j)) ? 1 : 0;
            }
            st.zfinal[(size_t)k * NF + rank] = v;
        }
    }
    if (lane == 0) {
        st.ray_active[k] = 0;
        st.iters[grp] = iter + 1;
    }
}

int check_cfg(const MpSamplerCfg* c) {
    if (c->n_samples_eval < 2 || c->n_samples_eval > 256 || c->n_samples < 1 || c->n_samples > 256 ||
        c->n_samples_extra < 0 || c->n_samples_extra > 256 || c->max_total_iters < 1 || c->max_total_iters > 8)
        return -1;
    return 0;
}

}  // namespace

extern "C" int mp_sampler_init(const MpSamplerCfg* cfg, const MpSamplerState* st, const float* far,
                               const int* hit_index, const int* hit_count, int max_rays, int group_size,
                               int n_rays_total, const float* t_rand, void* stream) {
    if (check_cfg(cfg)) return -1;
    if (max_rays <= 0) return 0;
    const int n_groups = (n_rays_total + group_size - 1) / group_size;
    int grid = (max_rays + WAVES - 1) / WAVES;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_sampler_init, dim3(grid), dim3(256), WAVES * cfg->n_samples_eval * 4, (hipStream_t)stream, *cfg,
                       *st, far, hit_index, hit_count, max_rays, n_groups, t_rand);
    return (int)hipGetLastError();
}

extern "C" int mp_sampler_bound(const MpSamplerCfg* cfg, const MpSamplerState* st, const float* beta0,
                                const int* hit_index, const int* hit_count, int max_rays, int group_size,
                                int n_rays_total, int iter, void* stream) {
    if (check_cfg(cfg)) return -1;
    if (max_rays <= 0) return 0;
    const int zm = cfg->n_samples_eval * cfg->max_total_iters;
    const int lds = WAVES * 5 * zm * 4;
    static int set_for = 0;
    if (lds > set_for) {
        hipFuncSetAttribute((const void*)k_sampler_bound, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        set_for = lds;
    }
    const int n_groups = (n_rays_total + group_size - 1) / group_size;
    hipLaunchKernelGGL(k_sampler_bound, dim3((max_rays + WAVES - 1) / WAVES), dim3(256), lds, (hipStream_t)stream, *cfg,
                       *st, beta0, hit_index, hit_count, max_rays, group_size, n_groups, iter, zm);
    return (int)hipGetLastError();
}

extern "C" int mp_sampler_resample(const MpSamplerCfg* cfg, const MpSamplerState* st, const float* beta0,
                                   const float* far, const int* hit_index, const int* hit_count, int max_rays,
                                   int group_size, int n_rays_total, int iter, const float* u_final,
                                   const int* extra_idx, void* stream) {
    if (check_cfg(cfg)) return -1;
    if (max_rays <= 0) return 0;
    const int zm = cfg->n_samples_eval * cfg->max_total_iters;
    const int lds = WAVES * 5 * zm * 4;
    static int set_for = 0;
    if (lds > set_for) {
        hipFuncSetAttribute((const void*)k_sampler_resample, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        set_for = lds;
    }
    const int n_groups = (n_rays_total + group_size - 1) / group_size;
    hipLaunchKernelGGL(k_sampler_resample, dim3((max_rays + WAVES - 1) / WAVES), dim3(256), lds, (hipStream_t)stream, *cfg,
                       *st, beta0, far, hit_index, hit_count, max_rays, group_size, n_groups, iter, zm, u_final,
                       extra_idx);
    return (int)hipGetLastError();
}
