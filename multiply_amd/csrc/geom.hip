// Geometry kernels: SMPL posing, nearest-vertex structure, canonical warp, rays and box culling.
// Entry points and the reference code they replace: include/multiply_hip.h.
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include "../../include/multiply_hip.h"
#include "common.hpp"

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int V = MP_SMPL_V, NJ = MP_SMPL_J, NC = MP_KNN_NC, CL = MP_KNN_CLUSTER;
constexpr int NCC = MP_KNN_NC / 2, CLC = 2 * MP_KNN_CLUSTER;      // the coarse granularity of the training searches: pairs of clusters (k_knn_build)
static_assert(MP_KNN_NC % 2 == 0 && 2 * MP_KNN_CLUSTER <= 64, "coarse clusters = pairs of fine ones, one wave each");
__host__ __device__ constexpr int mp_fine_clusters() { return MP_KNN_NC; }      // (where a local NC shadows the fine count)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// block-wide sum for blockDim.x = 256 (4 waves)
__device__ __forceinline__ float block_sum256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------------ SMPL (lbs.py)
// work layout (floats): v_shaped [3V] | J [72] | A [24*16] | pose_feature [207]
constexpr int W_VS = 0, W_J = 3 * V, W_A = W_J + 72 + 8, W_PF = W_A + NJ * 16;

__global__ void k_smpl_shape(const float* __restrict__ v_template, const float* __restrict__ shapedirs,
                             const float* __restrict__ params, float* __restrict__ work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over V*3
    if (i >= 3 * V) return;
    const float* betas = params + 76;
    float acc = 0.0f;
#pragma unroll
    for (int l = 0; l < 10; ++l) acc += betas[l] * shapedirs[(size_t)i * 10 + l];  // blend_shapes, lbs.py:252-273
    work[W_VS + i] = v_template[i] + acc;
}

__global__ __launch_bounds__(256) void k_smpl_joints(const float* __restrict__ j_regressor, float* __restrict__ work) {
    __shared__ float sh[4];
    const int j = blockIdx.x;  // joint
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) {  // vertices2joints, lbs.py:232-249
        const float w = j_regressor[(size_t)j * V + i];
        a0 += w * work[W_VS + 3 * i];
        a1 += w * work[W_VS + 3 * i + 1];
        a2 += w * work[W_VS + 3 * i + 2];
    }
    a0 = block_sum256(a0, sh);
    a1 = block_sum256(a1, sh);
    a2 = block_sum256(a2, sh);
    if (threadIdx.x == 0) { work[W_J + 3 * j] = a0; work[W_J + 3 * j + 1] = a1; work[W_J + 3 * j + 2] = a2; }
}

__device__ void mat4_mul(const float* a, const float* b, float* c) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}

__global__ __launch_bounds__(64) void k_smpl_chain(const int* __restrict__ parents, const float* __restrict__ params,
                                                   const float* __restrict__ tfs_c_inv, float* __restrict__ work,
                                                   float* __restrict__ tfs, float* __restrict__ joints) {
    __shared__ float R[NJ][9];
    __shared__ float G[NJ][16];
    const int t = threadIdx.x;
    const float scale = params[0];
    const float* transl = params + 1;
    const float* thetas = params + 4;
    const float* J = work + W_J;
    if (t < NJ) {  // batch_rodrigues, lbs.py:276-307
        const float rx0 = thetas[3 * t], ry0 = thetas[3 * t + 1], rz0 = thetas[3 * t + 2];
        const float ax = rx0 + 1e-8f, ay = ry0 + 1e-8f, az = rz0 + 1e-8f;
        const float angle = sqrtf(ax * ax + ay * ay + az * az);
        const float rx = rx0 / angle, ry = ry0 / angle, rz = rz0 / angle;
        float s, c;
        sincosf(angle, &s, &c);
        const float K[9] = {0.f, -rz, ry, rz, 0.f, -rx, -ry, rx, 0.f};
        float KK[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float a = 0.f;
                for (int k = 0; k < 3; ++k) a += K[3 * i + k] * K[3 * k + j];
                KK[3 * i + j] = a;
            }
        for (int i = 0; i < 9; ++i) R[t][i] = ((i % 4 == 0) ? 1.0f : 0.0f) + s * K[i] + (1.0f - c) * KK[i];
    }
    __syncthreads();
    // pose_feature = (R[1:] - I).flatten (lbs.py:199)
    for (int i = t; i < 207; i += 64) {
        const int j = i / 9 + 1, e = i % 9;
        work[W_PF + i] = R[j][e] - ((e % 4 == 0) ? 1.0f : 0.0f);
    }
    if (t == 0) {  // batch_rigid_transform, lbs.py:323-377 (24 tiny sequential 4x4 products)
        for (int j = 0; j < NJ; ++j) {
            const int p = parents[j];
            float rel[3];
            for (int a = 0; a < 3; ++a) rel[a] = J[3 * j + a] - (j > 0 ? J[3 * p + a] : 0.0f);
            float tm[16];
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) tm[4 * a + b] = R[j][3 * a + b];
                tm[4 * a + 3] = rel[a];
            }
            tm[12] = tm[13] = tm[14] = 0.f;
            tm[15] = 1.f;
            if (j == 0) for (int i = 0; i < 16; ++i) G[0][i] = tm[i];
            else mat4_mul(G[p], tm, G[j]);
        }
    }
    __syncthreads();
    if (t < NJ) {
        float A[16];
        for (int i = 0; i < 16; ++i) A[i] = G[t][i];
        // rel_transforms = G - pad(G @ [J;0])  (lbs.py:372-375)
        for (int a = 0; a < 4; ++a) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += G[t][4 * a + k] * J[3 * t + k];
            A[4 * a + 3] -= s;
        }
        for (int i = 0; i < 16; ++i) work[W_A + 16 * t + i] = A[i];
        // SMPLServer.forward scaling (smpl.py:80-91)
        float tf[16];
        for (int i = 0; i < 16; ++i) tf[i] = A[i];
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 4; ++b) tf[4 * a + b] *= scale;
            tf[4 * a + 3] += transl[a] * scale;
        }
        if (tfs_c_inv) {
            float o[16];
            mat4_mul(tf, tfs_c_inv + 16 * t, o);
            for (int i = 0; i < 16; ++i) tfs[16 * t + i] = o[i];
        } else {
            for (int i = 0; i < 16; ++i) tfs[16 * t + i] = tf[i];
        }
        for (int a = 0; a < 3; ++a) joints[3 * t + a] = G[t][4 * a + 3] * scale + transl[a] * scale;
    }
}

__global__ __launch_bounds__(256) void k_smpl_verts(const float* __restrict__ posedirs,
                                                    const float* __restrict__ lbs_weights,
                                                    const float* __restrict__ params, const float* __restrict__ work,
                                                    float* __restrict__ verts) {
    __shared__ float pf[207];
    __shared__ float A[NJ * 16];
    for (int i = threadIdx.x; i < 207; i += 256) pf[i] = work[W_PF + i];
    for (int i = threadIdx.x; i < NJ * 16; i += 256) A[i] = work[W_A + i];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc = 0.f;
        for (int q = 0; q < 207; ++q) acc += pf[q] * posedirs[(size_t)q * (3 * V) + 3 * v + k];  // lbs.py:201-202
        p[k] = acc + work[W_VS + 3 * v + k];
    }
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    for (int j = 0; j < NJ; ++j) {  // lbs.py:217-221
        const float w = lbs_weights[(size_t)v * NJ + j];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] += w * A[16 * j + i];
    }
    const float scale = params[0];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = T[4 * a] * p[0] + T[4 * a + 1] * p[1] + T[4 * a + 2] * p[2] + T[4 * a + 3];
        verts[3 * v + a] = x * scale + params[1 + a] * scale;  // smpl.py:77-78
    }
}

// ------------------------------------------------------------------------------------------------ KNN structure
// Blocks NC .. NC + NCC - 1 (round 6): the bounding sphere of the PAIR of clusters (2 cc, 2 cc + 1) -- consecutive kd leaves, siblings --
// as cbound[NC + cc].  The training searches (every sample of a ray needs its exact neighbour however far away it is, so many
// spheres are about equally near) run on these NCC = NC / 2 coarse clusters of 2 CL vertices: with the fine ones the training
// warp cost +30 % (twice the sphere tests and per-cluster steps), the eval searches -19 % (profiles/r06_cluster_ab.txt).
__global__ __launch_bounds__(64) void k_knn_build(const float* __restrict__ verts, const int* __restrict__ perm,
                                                  float4* __restrict__ vsorted, float4* __restrict__ cbound) {
    const int l = threadIdx.x;
    if (blockIdx.x >= NC) {
        const int cc = blockIdx.x - NC;
        const int id = l < 2 * CL ? perm[cc * 2 * CL + l] : -1;
        float x = 0.f, y = 0.f, z = 0.f;
        if (id >= 0) { x = verts[3 * id]; y = verts[3 * id + 1]; z = verts[3 * id + 2]; }
        const float n = fmaxf(wave_sum(id >= 0 ? 1.f : 0.f), 1.f);      // (an all-padding cluster: a zero sphere at the origin)
        const float cx = wave_sum(x) / n, cy = wave_sum(y) / n, cz = wave_sum(z) / n;
        const float dx = x - cx, dy = y - cy, dz = z - cz;
        const float r = wave_max(id >= 0 ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.f);
        if (l == 0) cbound[NC + cc] = make_float4(cx, cy, cz, r * 1.00001f + 1e-7f);
        return;
    }
    const int c = blockIdx.x;                           // one full wave per cluster; lanes >= CL (CL < 64) hold padding
    const int id = l < CL ? perm[c * CL + l] : -1;
    float x = 0.f, y = 0.f, z = 0.f;
    if (id >= 0) { x = verts[3 * id]; y = verts[3 * id + 1]; z = verts[3 * id + 2]; }
    const float n = fmaxf(wave_sum(id >= 0 ? 1.f : 0.f), 1.f);      // (an all-padding cluster: a zero sphere at the origin)
    const float cx = wave_sum(x) / n, cy = wave_sum(y) / n, cz = wave_sum(z) / n;
    const float dx = x - cx, dy = y - cy, dz = z - cz;
    const float r = wave_max(id >= 0 ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.f);
    float4 o;
    if (id >= 0) { o.x = x; o.y = y; o.z = z; o.w = __int_as_float(id); }
    else { o.x = 1e18f; o.y = 1e18f; o.z = 1e18f; o.w = __int_as_float(INT_MAX - 1); }
    if (l < CL) vsorted[c * CL + l] = o;
    if (l == 0) cbound[c] = make_float4(cx, cy, cz, r * 1.00001f + 1e-7f);
}

// -DMP_GEOM_PROF: cycle / event counters of the warp kernels, summed over waves (tools/geom_prof.py reads them):
//  [0] slabs  [1] cycles total  [2] cycles in knn cull (reductions + sphere tests)  [3] cycles in cluster scans
//  [4] clusters that passed the box cull  [5] clusters scanned  [6] cycles in loads  [7] cycles in the epilogue
#ifdef MP_GEOM_PROF
__device__ unsigned long long g_geom_prof[16];
#define GP_T() __builtin_readcyclecounter()
__shared__ unsigned long long gp_lds[16][16];   // per-wave accumulators: one global atomic per wave and counter at exit
#define GP_ADD(i, v) do { if ((threadIdx.x & 63) == 0) gp_lds[threadIdx.x >> 6][i] += (unsigned long long)(v); } while (0)
#define GP_BEGIN() do { if ((threadIdx.x & 63) < 16) gp_lds[threadIdx.x >> 6][threadIdx.x & 63] = 0; } while (0)
#define GP_END() do { if ((threadIdx.x & 63) < 16) atomicAdd(&g_geom_prof[threadIdx.x & 63], gp_lds[threadIdx.x >> 6][threadIdx.x & 63]); } while (0)
extern "C" int mp_geom_prof_read(unsigned long long* host16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_geom_prof), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_geom_prof), z, sizeof(z)); }
    return (int)e;
}
#else
#define GP_T() 0ull
#define GP_ADD(i, v) do { } while (0)
#define GP_BEGIN() do { } while (0)
#define GP_END() do { } while (0)
#endif

// Exact nearest vertex among the clustered set held in LDS (vs, cb), for the 64 points of one wave at once.
// The wave's points are spatially coherent (neighbouring rays at the same sample index), so clusters are culled ONCE per
// wave against the bounding box of its points: lane c tests clusters c and c+64 in parallel (two LDS reads instead of a
// latency-bound loop over all the spheres), and only the surviving clusters are scanned vertex by vertex.
//   cap2 (per lane): squared search radius; < 0 = idle lane.  A vertex within the cap, if any, is the exact nearest one
//   (ties -> lowest vertex id, like an argmin over the original order: pytorch3d knn_points / deformer.py:39).
//   Returns bi = INT_MAX when no vertex lies within the cap.
//   NC / CL: the granularity searched (the fine clusters, or pairs of them: cb = the coarse spheres, see k_knn_build).
#ifndef MP_KNN_BP
#define MP_KNN_BP 2      // vertex pairs read ahead in the cluster scan
#endif
template <int NCX = NC, int CLX = CL>
__device__ __forceinline__ void knn_capped(const float4* vs, const float4* cb, float px, float py, float pz, float cap2,
                                           float& best, int& bi) {
    constexpr int NC = NCX, CL = CLX;
    const int lane = threadIdx.x & 63;
    const bool on = cap2 >= 0.0f;
    const unsigned long long gp0 = GP_T();
    const float lx = wave_min(on ? px : FLT_MAX), ly = wave_min(on ? py : FLT_MAX), lz = wave_min(on ? pz : FLT_MAX);
    const float hx = wave_max(on ? px : -FLT_MAX), hy = wave_max(on ? py : -FLT_MAX), hz = wave_max(on ? pz : -FLT_MAX);
    const float capr = sqrtf(wave_max(on ? cap2 : 0.0f));
    constexpr int NH = (NC + 63) / 64;      // candidate masks: 64 clusters each
    unsigned long long cand[NH];
    // the candidate sphere closest to the middle of the wave's points is scanned first: it usually holds the neighbour of
    // most lanes, and with that distance in hand the per-cluster bound below rejects most of the other candidates
    const float mx = 0.5f * (lx + hx), my = 0.5f * (ly + hy), mz = 0.5f * (lz + hz);
    float nearest = FLT_MAX;
    int nearest_c = -1;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int c = lane + 64 * h;
        bool hit = false;
        if (c < NC) {
            const float4 b = cb[c];
            const float dx = b.x - fminf(fmaxf(b.x, lx), hx), dy = b.y - fminf(fmaxf(b.y, ly), hy),
                        dz = b.z - fminf(fmaxf(b.z, lz), hz);
            const float reach = (b.w + capr) * 1.0001f;
            hit = dx * dx + dy * dy + dz * dz <= reach * reach;
            const float ex = b.x - mx, ey = b.y - my, ez = b.z - mz;
            const float gap = sqrtf(ex * ex + ey * ey + ez * ez) - b.w;
            if (hit && gap < nearest) { nearest = gap; nearest_c = c; }
        }
        cand[h] = __ballot(hit);
    }
    int first_c = -1;
    unsigned long long any_c = 0;
#pragma unroll
    for (int h = 0; h < NH; ++h) any_c |= cand[h];
    if (any_c) {
        const float wm = wave_min(nearest);
        const unsigned long long who = __ballot(nearest_c >= 0 && nearest == wm);
        first_c = __shfl(nearest_c, __builtin_ctzll(who));
#pragma unroll
        for (int h = 0; h < NH; ++h)
            if (h == (first_c >> 6)) cand[h] &= ~(1ull << (first_c & 63));
    }
    // running minimum as ONE 64-bit key (distance bits << 32 | vertex id): distances are >= 0, so their bit patterns order
    // like the values, and a tie in distance falls through to the lower vertex id (the argmin order of the reference's
    // brute-force search) -- one v_cmp_lt_u64 and two selects per vertex, no branch in the scan.  Lanes that are off
    // hold key 0, which nothing undercuts.  [Round 6 tried (distance, id) with a FLOAT compare + a wave-wide tie mask that
    // repeats the search with these keys when a distance equals a running minimum: 14 instead of 16 VALU instructions per
    // vertex pair, and SLOWER (training warp 64 -> 79 us, sampler warp 2.24 -> 2.34 ms): the compares then write scalar
    // mask pairs (VOP3) that the selects read back behind wait states.  tests/test_geom_gpu.py holds the tie test it left.]
    unsigned long long key = on ? (((unsigned long long)__float_as_uint(cap2) << 32) | (unsigned)INT_MAX) : 0ull;
    const f32x2 PX = {px, px}, PY = {py, py}, PZ = {pz, pz};
    const unsigned long long gp1 = GP_T();
    GP_ADD(2, gp1 - gp0);
#pragma unroll
    for (int h = 0; h < NH; ++h) GP_ADD(4, __popcll(cand[h]));
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        unsigned long long m = cand[h];
        while (m || (h == 0 && first_c >= 0)) {
            int c;
            if (h == 0 && first_c >= 0) { c = first_c; first_c = -1; }
            else { c = __builtin_ctzll(m) + 64 * h; m &= m - 1; }
            const float4 b = cb[c];
            const float ex = px - b.x, ey = py - b.y, ez = pz - b.z;
            const float lb = fmaxf(sqrtf(ex * ex + ey * ey + ez * ez) - b.w, 0.0f);
            if (!__any(on && lb * lb * 0.9999f <= __uint_as_float((unsigned)(key >> 32)))) continue;   // no lane can improve here
            GP_ADD(5, 1);
            // two vertices per step in packed fp32 (v_pk_add / v_pk_mul / v_pk_fma_f32): the cluster is stored as pairs
            // (xa xb ya yb)(za zb ida idb), see load_knn_lds
            // The reads are software-pipelined by hand, KNN_BP pairs ahead (round 6): left to itself the compiler issued the two
            // reads of a pair and waited for them at once -- hidden by the other waves of an eval launch, but a training launch has
            // ONE wave per SIMD and paid the LDS latency 32 times per cluster (6.4 k cycles per 64 vertices, tools/geom_prof_train.py).
            const float4* cv = vs + c * CL;
            constexpr int BP = MP_KNN_BP;
            static_assert((CL / 2) % BP == 0, "pairs per block");
            float4 A[BP], B[BP], An[BP], Bn[BP];
#pragma unroll
            for (int j = 0; j < BP; ++j) { A[j] = cv[2 * j]; B[j] = cv[2 * j + 1]; }
#pragma unroll
            for (int k0 = 0; k0 < CL / 2; k0 += BP) {
                if (k0 + BP < CL / 2) {
#pragma unroll
                    for (int j = 0; j < BP; ++j) { An[j] = cv[2 * (k0 + BP + j)]; Bn[j] = cv[2 * (k0 + BP + j) + 1]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < BP; ++j) {
                    const f32x2 fx = PX - (f32x2){A[j].x, A[j].y}, fy = PY - (f32x2){A[j].z, A[j].w}, fz = PZ - (f32x2){B[j].x, B[j].y};
                    const f32x2 d2 = __builtin_elementwise_fma(fz, fz, __builtin_elementwise_fma(fy, fy, fx * fx));
                    const unsigned long long ka = ((unsigned long long)__float_as_uint(d2.x) << 32) | __float_as_uint(B[j].z);
                    const unsigned long long kb = ((unsigned long long)__float_as_uint(d2.y) << 32) | __float_as_uint(B[j].w);
                    key = ka < key ? ka : key;
                    key = kb < key ? kb : key;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < BP; ++j) { A[j] = An[j]; B[j] = Bn[j]; }
            }
        }
    }
    best = on ? __uint_as_float((unsigned)(key >> 32)) : cap2;
    bi = on ? (int)(unsigned)(key & 0xffffffffull) : INT_MAX;
    GP_ADD(3, GP_T() - gp1);
}

// Unbounded exact search.  An upper bound of the nearest-vertex distance is cheap: every cluster's bounding sphere
// contains at least one vertex, so d_nn <= min_c (|p - centre_c| + radius_c)  (NC broadcast LDS reads per lane).  With
// that per-lane cap a single culled pass finds the exact neighbour of near and far points alike (the previous scheme,
// growing a fixed cap geometrically, re-culled the clusters up to 7 times for the far samples of a training ray).
// want: lane participates.
template <bool NEAR_FIRST, int NCX = NC, int CLX = CL>
__device__ __forceinline__ void knn_unbounded(const float4* vs, const float4* cb, float px, float py, float pz, bool want,
                                              float& best, int& bi) {
    constexpr int NC = NCX, CL = CLX;
    best = -1.0f;
    bi = INT_MAX;
    bool todo = want;
    if constexpr (NEAR_FIRST) {   // canonical shading points sit near the surface: growing fixed radii resolve them fastest
        float cap = 0.0064f;      // (0.08)^2, then (0.16)^2, (0.32)^2
        for (int round = 0; round < 3 && __any(todo); ++round) {
            float b2; int i2;
            knn_capped<NC, CL>(vs, cb, px, py, pz, todo ? cap : -1.0f, b2, i2);
            if (todo && i2 != INT_MAX) { best = b2; bi = i2; todo = false; }
            cap *= 4.0f;
        }
    }
    if (__any(todo)) {
        float ub = FLT_MAX;
        for (int c = 0; c < NC; ++c) {
            const float4 b = cb[c];
            const float ex = px - b.x, ey = py - b.y, ez = pz - b.z;
            ub = fminf(ub, sqrtf(ex * ex + ey * ey + ez * ez) + b.w);
        }
        float b2; int i2;
        knn_capped<NC, CL>(vs, cb, px, py, pz, todo ? ub * ub * 1.0005f + 1e-12f : -1.0f, b2, i2);
        if (todo) { best = b2; bi = i2; }
    }
}

__device__ __forceinline__ void load_knn_lds(float4* vs, float4* cb, const float* vsorted, const float* cbound) {
    const float4* gv = (const float4*)vsorted;
    const float4* gc = (const float4*)cbound;
    for (int i = threadIdx.x; i < NC * CL / 2; i += blockDim.x) {      // vertex pair (2i, 2i+1) -> (xa xb ya yb)(za zb ida idb)
        const float4 a = gv[2 * i], b = gv[2 * i + 1];
        vs[2 * i] = make_float4(a.x, b.x, a.y, b.y);
        vs[2 * i + 1] = make_float4(a.z, b.z, a.w, b.w);
    }
    for (int i = threadIdx.x; i < NC; i += blockDim.x) cb[i] = gc[i];
}

// blended bone transform rows 0..2 (T[12]) and T[3][3] (s) of vertex `vid`
__device__ __forceinline__ void blend_tf(const float* __restrict__ skin_w, const float* tfs_lds, int vid, float (&T)[12],
                                         float& s33) {
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    s33 = 0.f;
    const float* w = skin_w + (size_t)vid * NJ;
    for (int j = 0; j < NJ; ++j) {
        const float wj = w[j];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] += wj * tfs_lds[16 * j + i];
        s33 += wj * tfs_lds[16 * j + 15];
    }
}

__device__ __forceinline__ void inv3(const float (&T)[12], float (&I)[9]) {
    const float a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], i = T[10];
    const float c0 = e * i - f * h, c1 = f * g - d * i, c2 = d * h - e * g;
    const float det = a * c0 + b * c1 + c * c2;
    const float r = 1.0f / det;
    I[0] = c0 * r; I[1] = (c * h - b * i) * r; I[2] = (b * f - c * e) * r;
    I[3] = c1 * r; I[4] = (a * i - c * g) * r; I[5] = (c * d - a * f) * r;
    I[6] = c2 * r; I[7] = (b * g - a * h) * r; I[8] = (a * e - b * d) * r;
}

// Per-vertex inverse blended transform of one pose: row r of vertex v = (I[3r], I[3r+1], I[3r+2], T[r][3] / T[3][3]) with
// T = sum_j w[v][j] tfs[j] and I = inverse of its 3x3 block.  The warp kernels map a point whose nearest vertex is v with
// x_c = I (x - c): one 48-byte gather per point instead of 24 scattered weight loads, 312 FMAs and a 3x3 inversion, with
// the same operations in the same order (the table is what every point with that neighbour computed for itself before).
__global__ void k_blend_table(const float* __restrict__ skin_w, const float* __restrict__ tfs, int n_verts,
                              float4* __restrict__ table) {
    __shared__ float tl[NJ * 16];
    for (int i = threadIdx.x; i < NJ * 16; i += blockDim.x) tl[i] = tfs[i];
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_verts) return;
    float T[12], s33, I[9];
    blend_tf(skin_w, tl, v, T, s33);
    inv3(T, I);
    table[3 * v] = make_float4(I[0], I[1], I[2], T[3] / s33);
    table[3 * v + 1] = make_float4(I[3], I[4], I[5], T[7] / s33);
    table[3 * v + 2] = make_float4(I[6], I[7], I[8], T[11] / s33);
}

constexpr int WARP_THREADS = 1024;
constexpr int WARP_LDS = NC * CL * 16 + (NC + NCC) * 16 + 32;      // vertices, fine + coarse spheres, box
// k_warp_inverse appends the ids of the points that need a network query to ONE list.  One returning atomicAdd per slab on that list's
// counter was the kernel: 624 k same-address atomics per frame serialise in one L2 channel (shading launches 4.75 ms, 2.30 ms with the
// atomic removed; profiles/r06_worklist_atomic.txt).  Every wave therefore stages ids in its own strip of LDS and reserves list space
// once per WL_STAGE - 64 ids or more (~8 slabs).
constexpr int WL_STAGE = 512;
constexpr int WARP_INV_LDS = WARP_LDS + (WARP_THREADS / 64) * WL_STAGE * 4;

// Which 64 (ray, sample) pairs share a wave of a rays-mode launch.  Rounds 1-5: 64 neighbouring rays x one sample index.  The per-point
// outputs are addressed [ray][sample], so every store instruction of such a wave touches 64 cache lines; the shading launch (mode 2: xc,
// flags, nearest-vertex index, and after it the Jacobian's 36 bytes per point) therefore takes MP_SLAB_RUN consecutive samples of 64 /
// MP_SLAB_RUN neighbouring rays (round 6) -- runs share their lines, the points stay as compact in space (pixels of a tile row x a short
// stretch of depth).  The sampler's launches (modes 0 / 1) write one value per point and measured slower that way: they keep 64 x 1.
#ifndef MP_SLAB_RUN
#define MP_SLAB_RUN 8
#endif
#ifndef MP_SLAB_RUN_SAMPLER
#define MP_SLAB_RUN_SAMPLER 1
#endif
__host__ __device__ constexpr int slab_run(int mode) { return mode == 2 ? MP_SLAB_RUN : MP_SLAB_RUN_SAMPLER; }

// one wave: reserve `staged` entries of the list and copy the wave's strip there (wave-private LDS: in order, no barrier)
__device__ __forceinline__ void worklist_flush(const int* stage, int staged, int* __restrict__ worklist, int* __restrict__ work_count, int lane) {
    int base = 0;
    if (lane == 0) base = atomicAdd(work_count, staged);
    base = __shfl(base, 0);
    for (int i = lane; i < staged; i += 64) worklist[base + i] = stage[i];
}

// mode 0: all points -> xc + worklist; mode 1: eval, outliers get sdf 4 and are skipped;
// mode 2: eval shading, outliers get sdf 4 and are skipped only when their alpha is exactly 0 in fp32
__global__ __launch_bounds__(WARP_THREADS) void k_warp_inverse(
    const float* __restrict__ pts, const float* __restrict__ dirs, const float* __restrict__ pose,
    const int* __restrict__ hit_index, const int* __restrict__ hit_count, const float* __restrict__ z, int z_stride,
    int n_s, int max_rays, int n_pts, const float* __restrict__ vsorted, const float* __restrict__ cbound,
    const float4* __restrict__ btab, int mode, const int* __restrict__ ray_active,
    const float* __restrict__ beta_p, const int* __restrict__ launch_active, float* __restrict__ xc,
    unsigned char* __restrict__ outlier, unsigned char* __restrict__ need_flag, float* __restrict__ sdf_out,
    int* __restrict__ worklist, int* __restrict__ work_count, int* __restrict__ nn_index,
    const float4* __restrict__ binned, const int* __restrict__ bincount) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (launch_active && *launch_active == 0) return;  // no ray of this launch is still being sampled
    float4* vs = (float4*)smem;
    float4* cb = vs + NC * CL;
    float4* cbc = cb + NC;           // the coarse spheres (training searches)
    float* box = (float*)(cbc + NCC);  // [6] conservative bounds of the vertex set (from the cluster spheres)
    int* stage = (int*)(smem + WARP_LDS) + (threadIdx.x >> 6) * WL_STAGE;   // this wave's strip of list entries not yet written out
    int staged = 0;
    load_knn_lds(vs, cb, vsorted, cbound);
    if (mode == 0) for (int i = threadIdx.x; i < NCC; i += blockDim.x) cbc[i] = ((const float4*)cbound)[NC + i];
    __syncthreads();
    if (threadIdx.x < 64) {
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int c = threadIdx.x; c < NC; c += 64) {
            const float4 b = cb[c];
            lo[0] = fminf(lo[0], b.x - b.w); lo[1] = fminf(lo[1], b.y - b.w); lo[2] = fminf(lo[2], b.z - b.w);
            hi[0] = fmaxf(hi[0], b.x + b.w); hi[1] = fmaxf(hi[1], b.y + b.w); hi[2] = fmaxf(hi[2], b.z + b.w);
        }
        for (int a = 0; a < 3; ++a) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); }
        if (threadIdx.x == 0) for (int a = 0; a < 3; ++a) { box[a] = lo[a] - 0.1005f; box[3 + a] = hi[a] + 0.1005f; }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    GP_BEGIN();
    const bool rays = pts == nullptr && binned == nullptr;
    const int n_rays = rays ? min(*hit_count, max_rays) : 0;
    if (binned) {          // the points in the order of k_warp_bin / k_warp_binned: n_pts = how many there are
        n_pts = 0;
        for (int c = 0; c < NCC; ++c) n_pts += bincount[c];
    }
    const int run = slab_run(mode), n_sr = (n_s + run - 1) / run, rpw = 64 / run;       // samples per run, runs per ray, rays per wave
    const int n_slab = rays ? ((n_rays + rpw - 1) / rpw) * n_sr : (n_pts + 63) / 64;
    float cam[3] = {0.f, 0.f, 0.f};
    if (rays) { cam[0] = pose[3]; cam[1] = pose[7]; cam[2] = pose[11]; }
    const float cap2 = 0.0101f;  // eval only needs neighbours within the 0.1 outlier radius
    for (int slab = blockIdx.x * nw + wave; slab < n_slab; slab += gridDim.x * nw) {
        int pid = -1;
        float x = 0.f, y = 0.f, zz = 0.f, dt = 0.f;
        const unsigned long long gs0 = GP_T();
        GP_ADD(0, 1);
        if (rays) {
            const int rb = slab / n_sr, k = rb * rpw + lane / run, s = (slab - rb * n_sr) * run + lane % run;
            if (k < n_rays && s < n_s && (!ray_active || ray_active[k])) {
                const int r = hit_index[k];
                const float t = z[(size_t)k * z_stride + s];
                x = cam[0] + t * dirs[3 * r]; y = cam[1] + t * dirs[3 * r + 1]; zz = cam[2] + t * dirs[3 * r + 2];
                pid = k * n_s + s;
                if (mode == 2) dt = z[(size_t)k * z_stride + s + 1] - t;
            }
        } else if (binned) {
            const int i = slab * 64 + lane;
            if (i < n_pts) { const float4 q = binned[i]; x = q.x; y = q.y; zz = q.z; pid = __float_as_int(q.w); }
        } else {
            const int i = slab * 64 + lane;
            if (i < n_pts) { x = pts[3 * i]; y = pts[3 * i + 1]; zz = pts[3 * i + 2]; pid = i; }
        }
        // eval: a point farther than 0.1 from the box of all vertices is an outlier without any search
        const bool near_box = mode == 0 || (x >= box[0] && y >= box[1] && zz >= box[2] && x <= box[3] && y <= box[4] &&
                                            zz <= box[5]);
        float best = -1.0f;
        int bi = INT_MAX;
#ifdef MP_GEOM_PROF
        x += __int_as_float(__float_as_int(x) & 0);   // keep the loads ahead of the stamp
        const unsigned long long gs1 = GP_T();
        GP_ADD(6, gs1 - gs0);
#endif
        if (mode == 0) knn_unbounded<false, NCC, CLC>(vs, cbc, x, y, zz, pid >= 0, best, bi);
        else if (__any(pid >= 0 && near_box))
            knn_capped(vs, cb, x, y, zz, (pid >= 0 && near_box) ? cap2 : -1.0f, best, bi);  // idle lanes open no cluster
        bool append = false, need_far = false, need = false, is_out = false;
        if (pid >= 0) {
            // outlier = sqrt(min(d2, 4)) > 0.1 (deformer.py:41-49)
            is_out = bi == INT_MAX || sqrtf(fminf(best, 4.0f)) > 0.1f;
            if (outlier) outlier[pid] = is_out ? 1 : 0;
            need = true;
            if (mode != 0 && is_out) {
                sdf_out[pid] = 4.0f;  // multiply.py:142-143
                need = false;
                if (mode == 2) {  // keep it only if its compositing weight can be non-zero
                    need = mp::alpha_of(4.0f, *beta_p, dt) != 0.0f;
                }
            }
            need_far = need && bi == INT_MAX;   // rare: outlier that still has weight -> exact unbounded search below
        }
        if (__any(need_far)) {
            float b2; int i2;
            knn_unbounded<false>(vs, cb, x, y, zz, need_far, b2, i2);
            if (need_far) { best = b2; bi = i2; }
        }
        const unsigned long long gs2 = GP_T();
        if (pid >= 0) {
            if (need_flag) need_flag[pid] = need ? 1 : 0;
            if (need) {
                const float4 r0 = btab[3 * bi], r1 = btab[3 * bi + 1], r2 = btab[3 * bi + 2];
                const float qx = x - r0.w, qy = y - r1.w, qz = zz - r2.w;
                xc[3 * (size_t)pid] = r0.x * qx + r0.y * qy + r0.z * qz;
                xc[3 * (size_t)pid + 1] = r1.x * qx + r1.y * qy + r1.z * qz;
                xc[3 * (size_t)pid + 2] = r2.x * qx + r2.y * qy + r2.z * qz;
                if (nn_index) nn_index[pid] = bi;
                append = worklist != nullptr;
            }
        }
        if (worklist) {
            const unsigned long long m = __ballot(append);
            if (m) {
                if (append) stage[staged + __popcll(m & ((1ull << lane) - 1ull))] = pid;
                staged += __popcll(m);
                if (staged > WL_STAGE - 64) { worklist_flush(stage, staged, worklist, work_count, lane); staged = 0; }
            }
        }
#ifdef MP_GEOM_PROF
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long gs3 = GP_T();
        GP_ADD(7, gs3 - gs2);
        GP_ADD(1, gs3 - gs0);
#endif
    }
    if (staged) worklist_flush(stage, staged, worklist, work_count, lane);
    GP_END();
}

// ---- TRAINING: the points of a wave grouped by their nearest vertex cluster -----------------------------------------------------
// A training batch is 512 RANDOM pixels: the 64 hit rays of a slab are scattered over the body's box, and most samples of a ray
// lie in free space far from the surface, where many clusters are about equally far.  The cluster scan is wave-uniform (a
// cluster is scanned when ANY lane may improve in it), so such a slab opened 50 of the (then 108) clusters (measured: 226 k cycles per
// slab, 1.15 ms per iteration).  Which 64 points share a wave is the kernel's own business -- results go out by point id -- so
// the points are first binned by the cluster whose bounding sphere is nearest (k_warp_bin: one atomic per point gives bin and
// rank), laid out bin after bin (k_warp_binned: position, point id), and k_warp_inverse walks that array.
__device__ __forceinline__ bool warp_sample_point(const float* dirs, const float* pose, const int* hit_index, const float* z,
                                                  int z_stride, int n_s, int n_rays, const int* ray_active, int i, float& x, float& y,
                                                  float& zz) {
    if (i >= n_rays * n_s) return false;
    const int k = i / n_s, s = i - k * n_s;
    if (ray_active && !ray_active[k]) return false;
    const int r = hit_index[k];
    const float t = z[(size_t)k * z_stride + s];
    x = pose[3] + t * dirs[3 * r]; y = pose[7] + t * dirs[3 * r + 1]; zz = pose[11] + t * dirs[3 * r + 2];
    return true;
}
__global__ __launch_bounds__(1024) void k_warp_bin(const float* __restrict__ dirs, const float* __restrict__ pose,
                                                   const int* __restrict__ hit_index, const int* __restrict__ hit_count,
                                                   const float* __restrict__ z, int z_stride, int n_s, int max_rays,
                                                   const float* __restrict__ cbound, const int* __restrict__ ray_active,
                                                   const int* __restrict__ launch_active, int* __restrict__ binrank,
                                                   int* __restrict__ bincount) {
    if (launch_active && *launch_active == 0) return;
    constexpr int NC = NCC;                    // bins = the coarse clusters the binned walk searches
    __shared__ float4 cb[NC];
    __shared__ int lcount[NC], lbase[NC];      // this workgroup's histogram, then its base rank in every bin
    for (int i = threadIdx.x; i < NC; i += blockDim.x) { cb[i] = ((const float4*)cbound)[mp_fine_clusters() + i]; lcount[i] = 0; }
    __syncthreads();
    const int n_rays = min(*hit_count, max_rays);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float x, y, zz;
    const bool on = warp_sample_point(dirs, pose, hit_index, z, z_stride, n_s, n_rays, ray_active, i, x, y, zz);
    int bc = 0, lrank = 0;
    if (on) {
        float best = FLT_MAX;
        for (int c = 0; c < NC; ++c) {
            const float4 b = cb[c];
            const float ex = x - b.x, ey = y - b.y, ez = zz - b.z;
            const float gap = sqrtf(ex * ex + ey * ey + ez * ez) - b.w;
            if (gap < best) { best = gap; bc = c; }
        }
        lrank = atomicAdd(&lcount[bc], 1);     // (LDS: the global counters see one add per workgroup and bin, not one per point --
    }                                          //  60 k same-address atomics on ~100 words took 110 us)
    __syncthreads();
    for (int c = threadIdx.x; c < NC; c += blockDim.x) lbase[c] = lcount[c] ? atomicAdd(&bincount[c], lcount[c]) : 0;
    __syncthreads();
    if (on) binrank[i] = (bc << 22) | (lbase[bc] + lrank);      // bin (< 512) | rank inside the bin (< 4 M points per launch); -1 = none
    else if (i < max_rays * n_s) binrank[i] = -1;
}
__global__ __launch_bounds__(256) void k_warp_binned(const float* __restrict__ dirs, const float* __restrict__ pose,
                                                     const int* __restrict__ hit_index, const int* __restrict__ hit_count,
                                                     const float* __restrict__ z, int z_stride, int n_s, int max_rays,
                                                     const int* __restrict__ launch_active, const int* __restrict__ binrank,
                                                     const int* __restrict__ bincount, float4* __restrict__ binned) {
    if (launch_active && *launch_active == 0) return;
    constexpr int NC = NCC;
    __shared__ int start[NC];
    if (threadIdx.x == 0) {
        int a = 0;
        for (int c = 0; c < NC; ++c) { start[c] = a; a += bincount[c]; }
    }
    __syncthreads();
    const int n_rays = min(*hit_count, max_rays);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * n_s) return;
    const int br = binrank[i];
    if (br < 0) return;
    float x, y, zz;
    warp_sample_point(dirs, pose, hit_index, z, z_stride, n_s, n_rays, nullptr, i, x, y, zz);
    binned[start[br >> 22] + (br & 0x3fffff)] = make_float4(x, y, zz, __int_as_float(i));
}

// Points are addressed like in k_warp_inverse (slab = 64 neighbouring hit rays x one sample index, so the 64 canonical
// points of a wave are spatially coherent); only points whose `need` flag is set are processed.
// Explicit-list variant (n_s == 0): point id = index, all `count` points processed.
__global__ __launch_bounds__(WARP_THREADS) void k_warp_jacobian(const float* __restrict__ xc,
                                                                const unsigned char* __restrict__ need,
                                                                const int* __restrict__ hit_count, int max_rays, int n_s,
                                                                int n_pts, const float* __restrict__ vsorted_c,
                                                                const float* __restrict__ cbound_c,
                                                                const float4* __restrict__ btab,
                                                                float* __restrict__ jinv,
                                                                int* __restrict__ nn_index,
                                                                const int* __restrict__ seed,
                                                                const float* __restrict__ verts_c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* vs = (float4*)smem;
    float4* cb = vs + NC * CL;
    load_knn_lds(vs, cb, vsorted_c, cbound_c);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    GP_BEGIN();
    const bool rays = n_s > 0;
    const int n_rays = rays ? min(*hit_count, max_rays) : 0;
    const int run = slab_run(2), n_sr = (n_s + run - 1) / run, rpw = 64 / run;          // (the shading launch's mapping, see k_warp_inverse)
    const int n_slab = rays ? ((n_rays + rpw - 1) / rpw) * n_sr : (n_pts + 63) / 64;
    for (int slab = blockIdx.x * nw + wave; slab < n_slab; slab += gridDim.x * nw) {
        int id = -1;
        if (rays) {
            const int rb = slab / n_sr, k = rb * rpw + lane / run, s = (slab - rb * n_sr) * run + lane % run;
            if (k < n_rays && s < n_s && need[(size_t)k * n_s + s]) id = k * n_s + s;
        } else {
            const int i = slab * 64 + lane;
            if (i < n_pts) id = i;
        }
        if (!__any(id >= 0)) continue;
        const unsigned long long gs0 = GP_T();
        GP_ADD(0, 1);
        float x = 0.f, y = 0.f, z = 0.f;
        if (id >= 0) { x = xc[3 * (size_t)id]; y = xc[3 * (size_t)id + 1]; z = xc[3 * (size_t)id + 2]; }
        float best; int bi;
        if (seed) {
            // the nearest POSED vertex of the deformed point is (almost always) also the nearest canonical vertex of x_c:
            // its canonical distance is a tight search radius, so one culled pass over very few clusters is exact
            float cap2 = -1.0f;
            if (id >= 0) {
                const int sv = seed[id];
                const float ex = x - verts_c[3 * sv], ey = y - verts_c[3 * sv + 1], ez = z - verts_c[3 * sv + 2];
                cap2 = (ex * ex + ey * ey + ez * ez) * 1.0005f + 1e-12f;
            }
            knn_capped(vs, cb, x, y, z, cap2, best, bi);
        } else {
            knn_unbounded<true>(vs, cb, x, y, z, id >= 0, best, bi);
        }
        if (id >= 0) {
            const float4 r0 = btab[3 * bi], r1 = btab[3 * bi + 1], r2 = btab[3 * bi + 2];
            float* o = jinv + 9 * (size_t)id;
            o[0] = r0.x; o[1] = r0.y; o[2] = r0.z; o[3] = r1.x; o[4] = r1.y; o[5] = r1.z; o[6] = r2.x; o[7] = r2.y; o[8] = r2.z;
            if (nn_index) nn_index[id] = bi;
        }
#ifdef MP_GEOM_PROF
        __builtin_amdgcn_s_waitcnt(0);
        GP_ADD(1, GP_T() - gs0);
#endif
    }
    GP_END();
}

// ------------------------------------------------------------------------------------------------ oriented box (PCA)
__device__ void jacobi3(float a[3][3], float v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.f : 0.f;
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabsf(a[p][q]) < 1e-20f) continue;
                const float th = (a[q][q] - a[p][p]) / (2.f * a[p][q]);
                const float t = (th >= 0.f ? 1.f : -1.f) / (fabsf(th) + sqrtf(th * th + 1.f));
                const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const float akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const float apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const float vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

__global__ __launch_bounds__(256) void k_obb(const float* __restrict__ verts, float inflate, float* __restrict__ obb) {
    __shared__ float sh[4];
    __shared__ float ax[9], mean[3];
    const int t = threadIdx.x;
    float m[3] = {0.f, 0.f, 0.f};
    for (int i = t; i < V; i += 256) { m[0] += verts[3 * i]; m[1] += verts[3 * i + 1]; m[2] += verts[3 * i + 2]; }
    for (int a = 0; a < 3; ++a) m[a] = block_sum256(m[a], sh) / V;
    float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < V; i += 256) {
        const float x = verts[3 * i] - m[0], y = verts[3 * i + 1] - m[1], z = verts[3 * i + 2] - m[2];
        cv[0] += x * x; cv[1] += x * y; cv[2] += x * z; cv[3] += y * y; cv[4] += y * z; cv[5] += z * z;
    }
    for (int a = 0; a < 6; ++a) cv[a] = block_sum256(cv[a], sh);
    if (t == 0) {
        float A[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}}, Vv[3][3];
        jacobi3(A, Vv);
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 3; ++k) ax[3 * a + k] = Vv[k][a];  // row a = eigenvector a
        for (int a = 0; a < 3; ++a) mean[a] = m[a];
    }
    __syncthreads();
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = t; i < V; i += 256) {
        const float x = verts[3 * i] - mean[0], y = verts[3 * i + 1] - mean[1], z = verts[3 * i + 2] - mean[2];
        for (int a = 0; a < 3; ++a) {
            const float p = ax[3 * a] * x + ax[3 * a + 1] * y + ax[3 * a + 2] * z;
            lo[a] = fminf(lo[a], p);
            hi[a] = fmaxf(hi[a], p);
        }
    }
    __shared__ float slo[4][3], shi[4][3];
    for (int a = 0; a < 3; ++a) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); }
    if ((t & 63) == 0) for (int a = 0; a < 3; ++a) { slo[t >> 6][a] = lo[a]; shi[t >> 6][a] = hi[a]; }
    __syncthreads();
    if (t == 0) {
        float mid[3], half[3];
        for (int a = 0; a < 3; ++a) {
            const float l = fminf(fminf(slo[0][a], slo[1][a]), fminf(slo[2][a], slo[3][a]));
            const float h = fmaxf(fmaxf(shi[0][a], shi[1][a]), fmaxf(shi[2][a], shi[3][a]));
            mid[a] = 0.5f * (l + h);
            half[a] = 0.5f * (h - l) * inflate;
        }
        for (int k = 0; k < 3; ++k) obb[k] = mean[k] + ax[k] * mid[0] + ax[3 + k] * mid[1] + ax[6 + k] * mid[2];
        for (int i = 0; i < 9; ++i) obb[3 + i] = ax[i];
        for (int a = 0; a < 3; ++a) obb[12 + a] = half[a];
    }
}

// ------------------------------------------------------------------------------------------------ rays
__global__ void k_ray_setup(const float* __restrict__ uv, const float* __restrict__ K, const float* __restrict__ P,
                            int n, float radius, float* __restrict__ dirs, float* __restrict__ far) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float fx = K[0], fy = K[5], cx = K[2], cy = K[6], sk = K[1];
    const float x = uv[2 * i], y = uv[2 * i + 1];
    // lift (rend_util.py:73-87) with z = 1
    const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx;
    const float yl = (y - cy) / fy;
    float w[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        w[a] = P[4 * a] * xl + P[4 * a + 1] * yl + P[4 * a + 2] + P[4 * a + 3];
        d[a] = w[a] - P[4 * a + 3];
    }
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);  // F.normalize
#pragma unroll
    for (int a = 0; a < 3; ++a) { d[a] /= nrm; dirs[3 * i + a] = d[a]; }
    // far root of the bounding sphere (rend_util.py:131-147)
    const float ox = P[3], oy = P[7], oz = P[11];
    const float b = d[0] * ox + d[1] * oy + d[2] * oz;
    const float under = b * b - ((ox * ox + oy * oy + oz * oz) - radius * radius);
    far[i] = fmaxf(sqrtf(under) - b, 0.0f);
}

__global__ void k_ray_box(const float* __restrict__ dirs, const float* __restrict__ P, const float* __restrict__ obb,
                          int n, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o[3] = {P[3] - obb[0], P[7] - obb[1], P[11] - obb[2]};
    const float d[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
    float tmin = -FLT_MAX, tmax = FLT_MAX;
    bool hit = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float* ax = obb + 3 + 3 * a;
        const float oo = ax[0] * o[0] + ax[1] * o[1] + ax[2] * o[2];
        const float dd = ax[0] * d[0] + ax[1] * d[1] + ax[2] * d[2];
        const float h = obb[12 + a];
        if (fabsf(dd) < 1e-12f) {
            hit = hit && fabsf(oo) <= h;
        } else {
            const float t0 = (-h - oo) / dd, t1 = (h - oo) / dd;
            tmin = fmaxf(tmin, fminf(t0, t1));
            tmax = fminf(tmax, fmaxf(t0, t1));
        }
    }
    flag[i] = (hit && tmax >= fmaxf(tmin, 0.0f)) ? 1 : 0;
}

// Eval-mode refinement of the box test, exact by construction: a ray that stays further than the outlier radius (0.1,
// deformer.py:49) from every vertex between `near` and its far end has only outlier samples, i.e. sdf = 4 on all of them
// (multiply.py:142-143); if moreover alpha = 1 - exp(-sigma(4) (far - near)) is exactly 0 in fp32 (it is for every beta
// below ~0.25: sigma(4) = e^(-4/beta) / (2 beta)) the ray's weights are exactly 0, its transmittance exactly 1, and its
// pixel is the background's -- bit for bit what a ray outside the box gets, and its beta converges in the first sampler
// iteration without touching its group's vote.  Such rays are dropped before they reach the sampler.  The test is
// conservative: the vertex set is covered by the cluster spheres (cbound), inflated by the radius plus a margin for the
// fp32 distance evaluation of the search kernels.
__global__ void k_ray_near_body(const float* __restrict__ dirs, const float* __restrict__ P, const float* __restrict__ cbound,
                                const float* __restrict__ far, const float* __restrict__ beta_p, float near_, int n,
                                int* __restrict__ flag) {
    __shared__ float4 cb[NC];
    for (int c = threadIdx.x; c < NC; c += blockDim.x) cb[c] = ((const float4*)cbound)[c];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const float tf = far[i];
    if (mp::alpha_of(4.0f, *beta_p, tf - near_) != 0.0f) return;   // outliers would still weigh in: keep the ray
    const float ox = P[3], oy = P[7], oz = P[11];
    const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    bool near_body = false;
    for (int c = 0; c < NC && !near_body; ++c) {
        const float4 b = cb[c];
        const float ex = b.x - ox, ey = b.y - oy, ez = b.z - oz;
        const float t = fminf(fmaxf(ex * dx + ey * dy + ez * dz, near_), tf);   // closest approach inside [near, far]
        const float qx = ex - t * dx, qy = ey - t * dy, qz = ez - t * dz;
        const float reach = b.w + 0.1005f;
        near_body = qx * qx + qy * qy + qz * qz <= reach * reach;
    }
    if (!near_body) flag[i] = 0;
}

// a convergence group without any hit gets its first ray (multiply.py:262-263 applied per group)
__global__ __launch_bounds__(256) void k_group_fallback(int* __restrict__ flag, int n, int group_size) {
    __shared__ int any;
    const int g0 = blockIdx.x * group_size;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    int a = 0;
    for (int i = g0 + threadIdx.x; i < min(n, g0 + group_size); i += 256) a |= flag[i];
    if (a) any = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !any) flag[g0] = 1;
}

constexpr int SCAN_BLOCK = 1024;
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_blocks(const int* __restrict__ flag, int n, int* __restrict__ bsum) {
    __shared__ int sh[SCAN_BLOCK / 64];
    const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int f = i < n ? flag[i] : 0;
    const int c = __popcll(__ballot(f != 0));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; ++w) s += sh[w];
        bsum[blockIdx.x] = s;
    }
}
__global__ void k_scan_top(int* __restrict__ bsum, int nb, int* __restrict__ total) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < nb; ++b) { const int c = bsum[b]; bsum[b] = s; s += c; }
        *total = s;
    }
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_scatter(const int* __restrict__ flag, int n,
                                                             const int* __restrict__ bsum, int* __restrict__ hit_index,
                                                             int* __restrict__ inv_index) {
    __shared__ int sh[SCAN_BLOCK / 64];
    const int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const int f = i < n ? flag[i] : 0;
    const unsigned long long m = __ballot(f != 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = __popcll(m);
    __syncthreads();
    int base = bsum[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += sh[w];
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (i < n) {
        inv_index[i] = f ? pos : -1;
        if (f) hit_index[pos] = i;
    }
}

__global__ void k_hits_from_index(const int* __restrict__ hit_index, int n_hit, int n_rays, int* __restrict__ hit_count,
                                  int* __restrict__ inv_index, int phase) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (phase == 0) {
        if (i < n_rays) inv_index[i] = -1;
        if (i == 0) *hit_count = n_hit;
    } else if (i < n_hit) {
        inv_index[hit_index[i]] = i;
    }
}

// ---- minimum-volume oriented box from the convex hull (multiply.py:208-214: trimesh's bounding_box_oriented) -------------
// The box is flush with a hull facet; on that facet's plane the minimum-area rectangle has a side along (the projection of) a
// SILHOUETTE edge of the hull (multiply_amd/obb.py, the published algorithm).  The hull comes from the host (Qhull, ~3 ms);
// the search -- facets x silhouette edges x hull vertices, ~10^7..10^8 fp64 operations that took the host 50 ms in numpy --
// runs here, one workgroup per facet normal, in the same order of preference as the host statement (first minimal edge per
// normal, first minimal normal), in double precision like it.
constexpr int OBB_T = 256;
__device__ __forceinline__ void obb_frame(const double* n, const double* e, double* d, double* w, bool& ok) {
    const double en = e[0] * n[0] + e[1] * n[1] + e[2] * n[2];
    d[0] = e[0] - en * n[0]; d[1] = e[1] - en * n[1]; d[2] = e[2] - en * n[2];
    const double ln = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    ok = ln > 1e-12;
    const double inv = ok ? 1.0 / ln : 0.0;
    d[0] *= inv; d[1] *= inv; d[2] *= inv;
    w[0] = n[1] * d[2] - n[2] * d[1]; w[1] = n[2] * d[0] - n[0] * d[2]; w[2] = n[0] * d[1] - n[1] * d[0];
}
// counts (optional, device): {hull vertices, facet normals, edges, status} written by k_hull_wrap -- the launch then covers the
// upper bound of facets and the blocks past the real count leave at once
__global__ __launch_bounds__(OBB_T) void k_obb_hull_search(const double* __restrict__ hv, int H, const double* __restrict__ normals,
                                                           const double* __restrict__ evec, const double* __restrict__ ena,
                                                           const double* __restrict__ enb, int E, double* __restrict__ work,
                                                           const int* __restrict__ counts, long long body_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sh = (double*)smem;                      // [H][3] hull vertices
    __shared__ double r_val[OBB_T];
    __shared__ int r_idx[OBB_T];
    __shared__ double r_lo[OBB_T], r_hi[OBB_T];
    {   // blockIdx.y = the body of a batch: its arrays lie body_stride doubles after the first body's, its counts 8 ints
        const long long o = (long long)blockIdx.y * body_stride;
        hv += o; normals += o; evec += o; ena += o; enb += o; work += o;
        if (counts) counts += 8 * blockIdx.y;
    }
    if (counts) {
        if ((int)blockIdx.x >= counts[1] || counts[3] != 0) return;
        H = counts[0];
        E = counts[2];
    }
    const int b = blockIdx.x, t = threadIdx.x;
    const double n[3] = {normals[3 * b], normals[3 * b + 1], normals[3 * b + 2]};
    for (int i = t; i < 3 * H; i += OBB_T) sh[i] = hv[i];
    __syncthreads();
    double lo = 1e300, hi = -1e300;
    for (int i = t; i < H; i += OBB_T) {
        const double h = sh[3 * i] * n[0] + sh[3 * i + 1] * n[1] + sh[3 * i + 2] * n[2];
        lo = fmin(lo, h); hi = fmax(hi, h);
    }
    double best = 1e300;
    int best_e = 0x7fffffff;
    for (int e = t; e < E; e += OBB_T) {
        const double sa = ena[3 * e] * n[0] + ena[3 * e + 1] * n[1] + ena[3 * e + 2] * n[2];
        const double sb = enb[3 * e] * n[0] + enb[3 * e + 1] * n[1] + enb[3 * e + 2] * n[2];
        if (!(sa * sb <= 1e-12)) continue;           // both facets face the same way: not on the silhouette
        double d[3], w[3];
        bool ok;
        obb_frame(n, evec + 3 * e, d, w, ok);
        if (!ok) continue;
        double ulo = 1e300, uhi = -1e300, wlo = 1e300, whi = -1e300;
        for (int i = 0; i < H; ++i) {
            const double x = sh[3 * i], y = sh[3 * i + 1], z = sh[3 * i + 2];
            const double pu = x * d[0] + y * d[1] + z * d[2], pw = x * w[0] + y * w[1] + z * w[2];
            ulo = fmin(ulo, pu); uhi = fmax(uhi, pu); wlo = fmin(wlo, pw); whi = fmax(whi, pw);
        }
        const double area = (uhi - ulo) * (whi - wlo);
        if (area < best) { best = area; best_e = e; }          // ascending e per thread: the first minimal edge wins a tie
    }
    r_val[t] = best; r_idx[t] = best_e; r_lo[t] = lo; r_hi[t] = hi;
    __syncthreads();
    for (int s = OBB_T / 2; s > 0; s >>= 1) {
        if (t < s) {
            if (r_val[t + s] < r_val[t] || (r_val[t + s] == r_val[t] && r_idx[t + s] < r_idx[t])) { r_val[t] = r_val[t + s]; r_idx[t] = r_idx[t + s]; }
            r_lo[t] = fmin(r_lo[t], r_lo[t + s]); r_hi[t] = fmax(r_hi[t], r_hi[t + s]);
        }
        __syncthreads();
    }
    if (t == 0) {
        work[2 * b] = r_idx[0] == 0x7fffffff ? 1e300 : r_val[0] * (r_hi[0] - r_lo[0]);   // volume of this facet's best box
        work[2 * b + 1] = (double)r_idx[0];
    }
}
__global__ __launch_bounds__(OBB_T) void k_obb_hull_pick(const double* __restrict__ hv, int H, const double* __restrict__ normals, int N,
                                                         const double* __restrict__ evec, const double* __restrict__ work,
                                                         float inflate, float* __restrict__ obb, const int* __restrict__ counts,
                                                         long long body_stride) {
    __shared__ double r_val[OBB_T];
    __shared__ int r_idx[OBB_T];
    __shared__ double r_lo[3][OBB_T], r_hi[3][OBB_T];
    const int t = threadIdx.x;
    {
        const long long o = (long long)blockIdx.x * body_stride;
        hv += o; normals += o; evec += o; work += o; obb += 16 * blockIdx.x;
        if (counts) counts += 8 * blockIdx.x;
    }
    if (counts) {
        H = counts[0];
        N = counts[3] != 0 ? 0 : counts[1];          // a failed hull: no candidate -> the all-zero record (the caller falls back)
    }
    double best = 1e300;
    int bi = 0x7fffffff;
    for (int b = t; b < N; b += OBB_T)
        if (work[2 * b] < best) { best = work[2 * b]; bi = b; }
    r_val[t] = best; r_idx[t] = bi;
    __syncthreads();
    for (int s = OBB_T / 2; s > 0; s >>= 1) {
        if (t < s && (r_val[t + s] < r_val[t] || (r_val[t + s] == r_val[t] && r_idx[t + s] < r_idx[t]))) { r_val[t] = r_val[t + s]; r_idx[t] = r_idx[t + s]; }
        __syncthreads();
    }
    const int b = r_idx[0];
    if (b == 0x7fffffff) {                            // degenerate hull: no candidate (never for a body)
        if (t < 16) obb[t] = 0.0f;
        return;
    }
    const int e = (int)work[2 * b + 1];
    double ax[3][3];
    ax[0][0] = normals[3 * b]; ax[0][1] = normals[3 * b + 1]; ax[0][2] = normals[3 * b + 2];
    bool ok;
    obb_frame(ax[0], evec + 3 * e, ax[1], ax[2], ok);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = t; i < H; i += OBB_T)
        for (int k = 0; k < 3; ++k) {
            const double p = hv[3 * i] * ax[k][0] + hv[3 * i + 1] * ax[k][1] + hv[3 * i + 2] * ax[k][2];
            lo[k] = fmin(lo[k], p); hi[k] = fmax(hi[k], p);
        }
    for (int k = 0; k < 3; ++k) { r_lo[k][t] = lo[k]; r_hi[k][t] = hi[k]; }
    __syncthreads();
    for (int s = OBB_T / 2; s > 0; s >>= 1) {
        if (t < s)
            for (int k = 0; k < 3; ++k) { r_lo[k][t] = fmin(r_lo[k][t], r_lo[k][t + s]); r_hi[k][t] = fmax(r_hi[k][t], r_hi[k][t + s]); }
        __syncthreads();
    }
    if (t == 0) {
        double c[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) {
            const double m = 0.5 * (r_lo[k][0] + r_hi[k][0]);
            for (int a = 0; a < 3; ++a) c[a] += m * ax[k][a];
        }
        for (int a = 0; a < 3; ++a) obb[a] = (float)c[a];
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) obb[3 + 3 * k + a] = (float)ax[k][a];
        for (int k = 0; k < 3; ++k) obb[12 + k] = (float)(0.5 * (r_hi[k][0] - r_lo[k][0]) * (double)inflate);
        obb[15] = 0.0f;
    }
}

// waves of work in a rays-mode launch (the kernels' own (ray, sample) -> wave mapping)
int ray_slabs(int max_rays, int n_s, int mode) {
    const int run = slab_run(mode), rpw = 64 / run;
    return ((max_rays + rpw - 1) / rpw) * ((n_s + run - 1) / run);
}

int warp_grid(int n_slab, int nw) {
    int g = (n_slab + nw - 1) / nw;
    return g < 1 ? 1 : (g > 256 ? 256 : g);
}

// Waves per workgroup of the warp kernels.  One workgroup per CU either way (its vertex structure fills 110 KB of LDS);
// 16 waves amortise that fill over a whole frame's slabs, but a training call has only ~900 slabs (512 rays x 128
// samples / 64) and would occupy 57 of the 256 CUs -- there, fewer waves per workgroup spread the slabs over the chip.
int warp_threads(int n_slab) {
    int nw = (n_slab + 255) / 256;          // waves per workgroup that give every CU a workgroup
    nw = nw < 1 ? 1 : (nw > WARP_THREADS / 64 ? WARP_THREADS / 64 : nw);
    return nw * 64;
}

}  // namespace

extern "C" int mp_smpl_pose(const float* v_template, const float* shapedirs, const float* posedirs,
                            const float* j_regressor, const float* lbs_weights, const int* parents, const float* params,
                            const float* tfs_c_inv, float* verts, float* tfs, float* joints, float* work, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_smpl_shape, dim3((3 * V + 255) / 256), dim3(256), 0, st, v_template, shapedirs, params, work);
    hipLaunchKernelGGL(k_smpl_joints, dim3(NJ), dim3(256), 0, st, j_regressor, work);
    hipLaunchKernelGGL(k_smpl_chain, dim3(1), dim3(64), 0, st, parents, params, tfs_c_inv, work, tfs, joints);
    hipLaunchKernelGGL(k_smpl_verts, dim3((V + 255) / 256), dim3(256), 0, st, posedirs, lbs_weights, params, work, verts);
    return (int)hipGetLastError();
}

extern "C" int mp_knn_build(const float* verts, const int* perm, float* vsorted, float* cbound, void* stream) {
    hipLaunchKernelGGL(k_knn_build, dim3(NC + NCC), dim3(64), 0, (hipStream_t)stream, verts, perm, (float4*)vsorted,
                       (float4*)cbound);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ convex hull on the device
// The hull the minimum-volume box search needs (facet normals, hull vertices, edges with the normals of their two facets) by GIFT
// WRAPPING: the reference's trimesh call (multiply.py:208-214) runs on the host behind a device -> host copy of the posed
// vertices, and round 3 kept that copy for Qhull.  A posed body's hull has 150-600 vertices / 300-1 200 facets.
//   pivot(a, b) = the vertex d with every other vertex q on the non-positive side of plane (a, b, d) (fp64 on the fp32
// coordinates: differences exact, products rounded).  All vertices lie within a half-turn around a hull edge, so the pivot is
// a reduction over an angle (hw_pivot_part).  The wrap is LEVEL-SYNCHRONOUS: the open edges of the current front are pivoted in
// parallel, one wave per edge -- or several waves per edge while the front is short -- by the HW_G workgroups of a body (see
// k_hull_wrap), then the master workgroup inserts the round's facets and collects the next front.  ~15 rounds, 0.37 ms for the
// bodies of a call.  [History: one workgroup pivoting one edge at a time, five barriers per facet: 2.1 ms per body; one
// workgroup, one wave per edge, pairwise orientation tests: 1.2 ms; the angle reduction alone changed nothing -- a single CU
// evaluates 700 pivots x 6 890 vertices whatever the predicate; spreading the pivots over 8 CUs did.]
// Ties (exactly coplanar vertices) go to the lower index; should they ever produce a non-manifold patch (a directed edge used
// twice) or the tables overflow, status is set and the caller falls back to the host-side hull.
// LDS per workgroup: the vertices (83 KB), an open-addressing table directed edge -> facet (48 KB, master only), the facets
// (12 KB), the front and the candidates' keys (12 KB).
constexpr int HW_T = 1024, HW_MAXF = 2048, HW_TAB = 8192, HW_MAXV = 6912, HW_FRONT = 1024;
constexpr int HW_LDS = HW_MAXV * 12 + HW_TAB * 4 + HW_TAB * 2 + HW_MAXF * 6 + 2 * HW_FRONT * 4 + HW_FRONT * 4 + 64 * 4 + 16 * 24;
// one WAVE: the pivot around the directed edge (v, u) away from a known supporting plane through it with OUTWARD normal n (the
// facet across the edge, or the start's virtual planes; n need not be normalised): with g = n x (u - v) -- in that plane,
// perpendicular to the edge, pointing away from the known facet -- every vertex q has w = q - v with s = -w . n >= 0, and the
// wrap's next vertex is the one whose half-plane through the edge makes the SMALLEST angle atan2(s, w . g) with g.  Angles in
// [0, pi] compare by cross-multiplication, c1 s2 - s1 c2 > 0, so a pivot is ONE branch-free pass (two fp64 dot products and a
// select per vertex) and a shuffle reduction of (c, s, index).  [The first version compared candidates pairwise with an
// orientation determinant and a plane that changed with the running best: every lane diverged, 38 k cycles per pivot.]
struct HwKey { double c, s; int i; };
__device__ __forceinline__ bool hw_key_better(const HwKey& cur, const HwKey& q) {      // does q beat cur?
    if (q.i < 0) return false;
    if (cur.i < 0) return true;
    const double x = q.c * cur.s - q.s * cur.c;
    if (x != 0.0) return x > 0.0;
    if (q.c * cur.c + q.s * cur.s < 0.0) return q.c > 0.0;       // opposite directions (angle 0 against pi)
    return q.i < cur.i;
}
// part / n_part: this wave scans vertices lane + 64 (part + n_part k) only (a pivot shared by n_part waves; hw_pivot_wave = all)
__device__ HwKey hw_pivot_part(const float* P, int V, int v, int u, double n0, double n1, double n2, int part, int n_part) {
    const int lane = threadIdx.x & 63;
    const double vx = P[3 * v], vy = P[3 * v + 1], vz = P[3 * v + 2];
    const double ex = (double)P[3 * u] - vx, ey = (double)P[3 * u + 1] - vy, ez = (double)P[3 * u + 2] - vz;
    const double g0 = n1 * ez - n2 * ey, g1 = n2 * ex - n0 * ez, g2 = n0 * ey - n1 * ex;       // n x (u - v)
    HwKey best = {0.0, 0.0, -1};
    const int step = 64 * n_part;
    for (int q0 = lane + 64 * part; q0 < V; q0 += 2 * step) {       // two vertices per trip: independent chains
        HwKey k[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + j * step, qc = min(q, V - 1);
            const double wx = (double)P[3 * qc] - vx, wy = (double)P[3 * qc + 1] - vy, wz = (double)P[3 * qc + 2] - vz;
            k[j].c = wx * g0 + wy * g1 + wz * g2;
            k[j].s = fmax(-(wx * n0 + wy * n1 + wz * n2), 0.0);                               // s < 0 is rounding only
            // not: beyond the end, the edge's own vertices, points on its line
            k[j].i = (q >= V || q == v || q == u || (k[j].c == 0.0 && k[j].s == 0.0)) ? -1 : q;
        }
        if (hw_key_better(k[0], k[1])) k[0] = k[1];
        if (hw_key_better(best, k[0])) best = k[0];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        HwKey other;
        other.c = __shfl_xor(best.c, o);
        other.s = __shfl_xor(best.s, o);
        other.i = __shfl_xor(best.i, o);
        if (hw_key_better(best, other)) best = other;
    }
    return best;
}
__device__ __forceinline__ int hw_pivot_wave(const float* P, int V, int v, int u, double n0, double n1, double n2) {
    return hw_pivot_part(P, V, v, u, n0, n1, n2, 0, 1).i;
}
// outward (unnormalised) normal of facet f
__device__ __forceinline__ void hw_facet_normal(const float* P, const unsigned short* fac, int f, double& n0, double& n1, double& n2) {
    const int a = fac[3 * f], b = fac[3 * f + 1], c = fac[3 * f + 2];
    const double ux = (double)P[3 * b] - P[3 * a], uy = (double)P[3 * b + 1] - P[3 * a + 1], uz = (double)P[3 * b + 2] - P[3 * a + 2];
    const double wx = (double)P[3 * c] - P[3 * a], wy = (double)P[3 * c + 1] - P[3 * a + 1], wz = (double)P[3 * c + 2] - P[3 * a + 2];
    n0 = uy * wz - uz * wy; n1 = uz * wx - ux * wz; n2 = ux * wy - uy * wx;
}
__device__ __forceinline__ unsigned hw_slot(unsigned key) { return (key * 2654435761u) >> 19; }   // 13 bits
// directed edge (u, v) -> facet id, or -1
__device__ int hw_find(const unsigned* keys, const unsigned short* vals, int u, int v) {
    const unsigned key = ((unsigned)u << 16) | (unsigned)v | 0x80000000u;
    for (unsigned s = hw_slot(key), n = 0; n < HW_TAB; s = (s + 1) & (HW_TAB - 1), ++n) {
        if (keys[s] == key) return vals[s];
        if (keys[s] == 0u) return -1;
    }
    return -1;
}
__device__ bool hw_insert(unsigned* keys, unsigned short* vals, int u, int v, int f) {
    const unsigned key = ((unsigned)u << 16) | (unsigned)v | 0x80000000u;
    for (unsigned s = hw_slot(key), n = 0; n < HW_TAB; s = (s + 1) & (HW_TAB - 1), ++n) {
        if (keys[s] == key) return false;                 // the directed edge exists already: not a 2-manifold
        if (keys[s] == 0u) { keys[s] = key; vals[s] = (unsigned short)f; return true; }
    }
    return false;
}
// ---- the wrap across HW_G workgroups: every one holds the vertices in its LDS and pivots a share of the front's edges (one
// wave per edge, edges dealt across workgroups first so that a wave has its SIMD to itself while the front is short); the
// MASTER workgroup alone keeps the edge table and the facets, inserts a round's facets IN PARALLEL (duplicates -- a triangle
// reached from two or three of its edges -- found by comparing canonical keys, ids by a prefix sum, table slots claimed with
// LDS compare-and-swap) and publishes the next front.  Two grid barriers per round on a counter in global memory; the exchanged
// words (front records, pivots) go through agent-scope atomics.  Consecutive workgroup ids go round the 8 XCDs: the HW_G
// workers of a body are the workgroups 8 j + x of ONE x, so they share one XCD's L2 and the barrier stays inside it; the bodies
// of a batch take different XCDs (body % 8), workgroups without a body leave at once.
constexpr int HW_G = 8;
__device__ __forceinline__ void hw_store(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned hw_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// false: the wrap was abandoned -- a peer waited ~30 ms for this barrier (xch[3]; round 4 waited ~0.3 s: a stall that long per
// iteration is worse than the fall-back it avoids).  The HW_G workgroups of a body spin on each
// other, so they must all be resident; should something else hold the XCD's CUs for good (several processes sharing the GPU,
// each with a partly scheduled wrap), the kernel gives up instead of hanging and the caller takes the host-side hull.
__device__ __forceinline__ bool hw_grid_barrier(unsigned* xch, unsigned& target, int* lds_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += HW_G;
        __hip_atomic_fetch_add(xch, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        for (unsigned spins = 0; __hip_atomic_load(xch, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
            if (hw_load(&xch[3]) != 0u) { ok = 0; break; }
            if (spins > (1u << 15)) { hw_store(&xch[3], 1u); ok = 0; break; }   // ~30 ms; a round's barrier normally takes ~10 us
            __builtin_amdgcn_s_sleep(1);
        }
        *lds_flag = ok;
    }
    __syncthreads();
    return *lds_flag != 0;
}
// exclusive prefix sum of one small count per thread over the workgroup (two barriers); tot = the sum
__device__ __forceinline__ int hw_scan(int x, int* wsum, int& tot) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(inc, o); if (lane >= o) inc += y; }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0; tot = 0;
    for (int w = 0; w < HW_T / 64; ++w) { const int y = wsum[w]; if (w < wave) base += y; tot += y; }
    return base + inc - x;
}
__device__ __forceinline__ bool hw_insert_cas(unsigned* keys, unsigned short* vals, int u, int v, int f) {
    const unsigned key = ((unsigned)u << 16) | (unsigned)v | 0x80000000u;
    for (unsigned s = hw_slot(key), n = 0; n < HW_TAB; s = (s + 1) & (HW_TAB - 1), ++n) {
        const unsigned old = atomicCAS(&keys[s], 0u, key);
        if (old == 0u) { vals[s] = (unsigned short)f; return true; }
        if (old == key) return false;                     // the directed edge exists already: not a 2-manifold
    }
    return false;
}
// xch (global, zeroed by the launcher): [0] barrier counter, [1] front size, [2] failed, [3] abandoned; records [HW_FRONT][4] at word 64:
// {u << 16 | v, a << 16 | b, c, -} = the open edge and the facet it belongs to; pivots [HW_FRONT] after them.
// out: counts {H, F, E, status, rounds, clocks}; hv [<= V][3], normals [<= HW_MAXF][3], evec / ena / enb [<= 3 HW_MAXF / 2][3]  (fp64)
__global__ __launch_bounds__(HW_T) void k_hull_wrap(const float* __restrict__ verts, int V, unsigned* __restrict__ xch,
                                                    int* __restrict__ counts, double* __restrict__ hv, double* __restrict__ normals,
                                                    double* __restrict__ evec, double* __restrict__ ena, double* __restrict__ enb,
                                                    long long body_stride, int n_bodies) {
    const int body = 8 * ((blockIdx.x >> 3) / HW_G) + (blockIdx.x & 7), wg = (blockIdx.x >> 3) % HW_G;
    if (body >= n_bodies) return;
    const bool master = wg == 0;
    {   // this body's vertices, exchange area, outputs
        const long long o = (long long)body * body_stride;
        verts += (long long)body * 3 * V; xch += 2 * o; counts += 8 * body;
        hv += o; normals += o; evec += o; ena += o; enb += o;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* P = (float*)smem;
    unsigned* keys = (unsigned*)(smem + HW_MAXV * 12);
    unsigned short* vals = (unsigned short*)(smem + HW_MAXV * 12 + HW_TAB * 4);
    unsigned short* fac = (unsigned short*)(smem + HW_MAXV * 12 + HW_TAB * 6);
    unsigned* front = (unsigned*)(smem + HW_MAXV * 12 + HW_TAB * 6 + HW_MAXF * 6);            // [HW_FRONT]: (u << 16) | v
    unsigned* ck0 = front + HW_FRONT;                                                         // [HW_FRONT] canonical triangle keys
    int* cand = (int*)(front + 2 * HW_FRONT);                                                 // [HW_FRONT]
    int* red = cand + HW_FRONT;                            // [0,16) wave results, [32..] control words
    HwKey* pk = (HwKey*)(red + 64);                        // [16] the waves' partial pivots
    unsigned* grec = xch + 64;
    unsigned* gcand = xch + 64 + 4 * HW_FRONT;
    const int t = threadIdx.x, wave = t >> 6;
    unsigned bar_target = 0;
    for (int i = t; i < 3 * V; i += HW_T) P[i] = verts[i];
    if (master) for (int i = t; i < HW_TAB; i += HW_T) keys[i] = 0u;
    __syncthreads();
    // ---- the first facet (master): lowest x (ties: lowest index); pivot around the vertical line through it; pivot around that edge
    if (master) {
        int best = -1;
        for (int q = t; q < V; q += HW_T)
            if (best < 0 || P[3 * q] < P[3 * best] || (P[3 * q] == P[3 * best] && q < best)) best = q;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int other = __shfl_xor(best, o);
            if (other >= 0 && (best < 0 || P[3 * other] < P[3 * best] || (P[3 * other] == P[3 * best] && other < best))) best = other;
        }
        if ((t & 63) == 0) red[wave] = best;
        __syncthreads();
        if (t == 0) {
            int b = red[0];
            for (int w = 1; w < HW_T / 64; ++w) {
                const int other = red[w];
                if (other >= 0 && (b < 0 || P[3 * other] < P[3 * b] || (P[3 * other] == P[3 * b] && other < b))) b = other;
            }
            red[32] = b;
            // a virtual vertex straight above p0 (slot V of the coordinate array: HW_MAXV > V is guaranteed by the launcher)
            P[3 * V] = P[3 * b]; P[3 * V + 1] = P[3 * b + 1]; P[3 * V + 2] = P[3 * b + 2] + 1.0f;
        }
        __syncthreads();
        if (wave == 0) {
            const int p0 = red[32];
            // the plane x = x(p0) supports the hull (outward normal -x): pivot around the vertical line through p0 -> a hull edge
            const int p1 = hw_pivot_wave(P, V, p0, V, -1.0, 0.0, 0.0);
            // the plane through that edge and the vertical supports the hull too (every vertex has w . ((p1 - p0) x z) >= 0):
            // pivot around (p0, p1) -> the first facet, every vertex on its non-positive side
            int p2 = -1;
            if (p1 >= 0) {
                const double dx = (double)P[3 * p1] - P[3 * p0], dy = (double)P[3 * p1 + 1] - P[3 * p0 + 1];
                p2 = hw_pivot_wave(P, V, p0, p1, -dy, dx, 0.0);
            }
            if (t == 0) {
                const bool ok = p1 >= 0 && p2 >= 0;
                red[33] = ok ? 0 : 1;                      // failed
                red[34] = 1;                               // F
                red[35] = 0;                               // front size
                if (ok) {
                    fac[0] = (unsigned short)p0; fac[1] = (unsigned short)p1; fac[2] = (unsigned short)p2;
                    hw_insert(keys, vals, p0, p1, 0); hw_insert(keys, vals, p1, p2, 0); hw_insert(keys, vals, p2, p0, 0);
                    const int tri[4] = {p0, p1, p2, p0};
                    for (int k = 0; k < 3; ++k) {
                        front[k] = ((unsigned)tri[k] << 16) | (unsigned)tri[k + 1];
                        hw_store(&grec[4 * k], front[k]);
                        hw_store(&grec[4 * k + 1], ((unsigned)p0 << 16) | (unsigned)p1);
                        hw_store(&grec[4 * k + 2], (unsigned)p2);
                    }
                    red[35] = 3;
                }
                hw_store(&xch[1], (unsigned)red[35]);
                hw_store(&xch[2], (unsigned)red[33]);
            }
        }
    }
    // ---- wrap, one round per front: edge (u, v) of a facet has its twin (v, u) in the facet across it
    long long t_piv = 0, t_ins = 0, t_all = clock64();
    int n_round = 0;
    bool abandoned = false;
    for (int round = 0; round < 4 * HW_MAXF; ++round) {
        if (!hw_grid_barrier(xch, bar_target, red + 39)) { abandoned = true; break; }      // the front is published
        const int n = (int)hw_load(&xch[1]);
        if (n == 0 || hw_load(&xch[2]) != 0u) break;
        ++n_round;
        const long long t0 = clock64();
        // this workgroup's edges are wg, wg + HW_G, ...; while there are fewer of them than waves, n_part waves share one pivot
        const int n_wg = (n - wg + HW_G - 1) / HW_G;
        int n_part = 1;
        while (n_part < HW_T / 64 && 2 * n_part * n_wg <= HW_T / 64) n_part *= 2;
        const int slots = (HW_T / 64) / n_part, slot = wave / n_part, part = wave % n_part;
        for (int e0 = 0; e0 < n_wg; e0 += slots) {
            const int e = wg + HW_G * (e0 + slot);
            const bool has = e0 + slot < n_wg;
            HwKey k = {0.0, 0.0, -1};
            if (has) {
                const unsigned r0 = hw_load(&grec[4 * e]), r1 = hw_load(&grec[4 * e + 1]), r2 = hw_load(&grec[4 * e + 2]);
                const int u = (int)(r0 >> 16), v = (int)(r0 & 0xffffu);
                const int a = (int)(r1 >> 16), b = (int)(r1 & 0xffffu), c = (int)r2;
                const double ux = (double)P[3 * b] - P[3 * a], uy = (double)P[3 * b + 1] - P[3 * a + 1], uz = (double)P[3 * b + 2] - P[3 * a + 2];
                const double wx = (double)P[3 * c] - P[3 * a], wy = (double)P[3 * c + 1] - P[3 * a + 1], wz = (double)P[3 * c + 2] - P[3 * a + 2];
                k = hw_pivot_part(P, V, v, u, uy * wz - uz * wy, uz * wx - ux * wz, ux * wy - uy * wx, part, n_part);
            }
            if (n_part == 1) {
                if (has && (t & 63) == 0) hw_store(&gcand[e], (unsigned)k.i);
                continue;
            }
            if ((t & 63) == 0) { pk[wave].c = k.c; pk[wave].s = k.s; pk[wave].i = k.i; }
            __syncthreads();
            if (has && part == 0 && (t & 63) == 0) {
                for (int j = 1; j < n_part; ++j) if (hw_key_better(k, pk[wave + j])) k = pk[wave + j];
                hw_store(&gcand[e], (unsigned)k.i);
            }
            __syncthreads();
        }
        if (!hw_grid_barrier(xch, bar_target, red + 39)) { abandoned = true; break; }      // the pivots are published
        const long long t1 = clock64();
        t_piv += t1 - t0;
        if (!master) continue;
        // ---- insert (n <= HW_FRONT = HW_T: one candidate per thread)
        int F = red[34];
        int u = 0, v = 0, d = -1;
        bool mine = false;
        if (t < n) {
            u = (int)(front[t] >> 16); v = (int)(front[t] & 0xffffu); d = (int)hw_load(&gcand[t]);
            if (d < 0) red[33] = 1;
            // the triangle (v, u, d) rotated to start at its lowest vertex
            int a = v, b = u, c = d;
            if (b < a && b < c) { a = u; b = d; c = v; } else if (c < a && c < b) { a = d; b = v; c = u; }
            ck0[t] = ((unsigned)a << 16) | (unsigned)b; cand[t] = c;
        }
        __syncthreads();
        if (t < n && d >= 0) {
            mine = true;
            const unsigned k0 = ck0[t]; const int k1 = cand[t];
            for (int e = 0; e < t; ++e) if (ck0[e] == k0 && cand[e] == k1) { mine = false; break; }
        }
        int n_new;
        const int f = F + hw_scan(mine ? 1 : 0, red, n_new);
        if (F + n_new > HW_MAXF) { if (t == 0) red[33] = 1; }
        else if (mine) {
            fac[3 * f] = (unsigned short)v; fac[3 * f + 1] = (unsigned short)u; fac[3 * f + 2] = (unsigned short)d;
            if (!(hw_insert_cas(keys, vals, v, u, f) & hw_insert_cas(keys, vals, u, d, f) & hw_insert_cas(keys, vals, d, v, f))) red[33] = 1;
        }
        __syncthreads();
        // the new facets' two other edges are open unless their twins exist (now: every facet of the round is in the table)
        const bool failed = red[33] != 0;
        const bool o0 = mine && !failed && hw_find(keys, vals, d, u) < 0;      // edge (u, d)
        const bool o1 = mine && !failed && hw_find(keys, vals, v, d) < 0;      // edge (d, v)
        int m;
        int at = hw_scan((o0 ? 1 : 0) + (o1 ? 1 : 0), red, m);
        __syncthreads();                                   // (front[] was read above; it is rewritten below)
        if (m > HW_FRONT) { if (t == 0) red[33] = 1; m = 0; }
        else {
            const unsigned fa = ((unsigned)v << 16) | (unsigned)u;
            if (o0) { front[at] = ((unsigned)u << 16) | (unsigned)d; hw_store(&grec[4 * at], front[at]); hw_store(&grec[4 * at + 1], fa);
                      hw_store(&grec[4 * at + 2], (unsigned)d); ++at; }
            if (o1) { front[at] = ((unsigned)d << 16) | (unsigned)v; hw_store(&grec[4 * at], front[at]); hw_store(&grec[4 * at + 1], fa);
                      hw_store(&grec[4 * at + 2], (unsigned)d); }
        }
        __syncthreads();
        if (t == 0) {
            red[34] = F + n_new; red[35] = m;
            hw_store(&xch[1], red[33] != 0 ? 0u : (unsigned)m);
            hw_store(&xch[2], (unsigned)red[33]);
        }
        t_ins += clock64() - t1;
    }
    if (!master) return;
    if (t == 0) { counts[4] = n_round; counts[5] = (int)(t_piv >> 4); counts[6] = (int)(t_ins >> 4); counts[7] = (int)((clock64() - t_all) >> 4); }
    const int F = red[34];
    const bool fail = abandoned || red[33] != 0 || red[35] != 0;
    __syncthreads();
    // ---- outputs
    int* cnt = red + 40;                                   // [0] hull vertices, [1] edges
    if (t == 0) { cnt[0] = 0; cnt[1] = 0; }
    unsigned* used = keys;                                 // (the table is read below: the marks go to the facet-normal pass first)
    __syncthreads();
    if (fail) {
        if (t == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 0; counts[3] = 1; }
        return;
    }
    for (int f = t; f < F; f += HW_T) {                    // outward unit normals; the search's copy with the reference's sign rule
        const int a = fac[3 * f], b = fac[3 * f + 1], c = fac[3 * f + 2];
        const double ux = (double)P[3 * b] - P[3 * a], uy = (double)P[3 * b + 1] - P[3 * a + 1], uz = (double)P[3 * b + 2] - P[3 * a + 2];
        const double vx = (double)P[3 * c] - P[3 * a], vy = (double)P[3 * c + 1] - P[3 * a + 1], vz = (double)P[3 * c + 2] - P[3 * a + 2];
        double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
        const double ln = sqrt(nx * nx + ny * ny + nz * nz), inv = ln > 0.0 ? 1.0 / ln : 0.0;
        nx *= inv; ny *= inv; nz *= inv;
        const bool flip = nx < 0.0 || (nx == 0.0 && ny < 0.0) || (nx == 0.0 && ny == 0.0 && nz < 0.0);     // obb.py _hull_parts
        normals[3 * f] = flip ? -nx : nx; normals[3 * f + 1] = flip ? -ny : ny; normals[3 * f + 2] = flip ? -nz : nz;
    }
    for (int f = t; f < F; f += HW_T)                      // edges (u < v) with the OUTWARD normals of their two facets
        for (int k = 0; k < 3; ++k) {
            const int u = fac[3 * f + k], v = fac[3 * f + (k + 1) % 3];
            if (u > v) continue;
            const int g = hw_find(keys, vals, v, u);
            const int e = atomicAdd(&cnt[1], 1);
            for (int a = 0; a < 3; ++a) evec[3 * e + a] = (double)P[3 * v + a] - (double)P[3 * u + a];
            for (int side = 0; side < 2; ++side) {
                const int ff = side == 0 ? f : g;
                const int a = fac[3 * ff], b = fac[3 * ff + 1], c = fac[3 * ff + 2];
                const double ux = (double)P[3 * b] - P[3 * a], uy = (double)P[3 * b + 1] - P[3 * a + 1], uz = (double)P[3 * b + 2] - P[3 * a + 2];
                const double vx = (double)P[3 * c] - P[3 * a], vy = (double)P[3 * c + 1] - P[3 * a + 1], vz = (double)P[3 * c + 2] - P[3 * a + 2];
                double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
                const double ln = sqrt(nx * nx + ny * ny + nz * nz), inv = ln > 0.0 ? 1.0 / ln : 0.0;
                double* o = side == 0 ? ena : enb;
                o[3 * e] = nx * inv; o[3 * e + 1] = ny * inv; o[3 * e + 2] = nz * inv;
            }
        }
    __syncthreads();
    // hull vertices: marks in the (now idle) key table region
    for (int i = t; i < V; i += HW_T) used[i] = 0u;
    __syncthreads();
    for (int i = t; i < 3 * F; i += HW_T) used[fac[i]] = 1u;
    __syncthreads();
    for (int i = t; i < V; i += HW_T)
        if (used[i]) {
            const int h = atomicAdd(&cnt[0], 1);
            hv[3 * h] = P[3 * i]; hv[3 * h + 1] = P[3 * i + 1]; hv[3 * h + 2] = P[3 * i + 2];
        }
    __syncthreads();
    if (t == 0) { counts[0] = cnt[0]; counts[1] = F; counts[2] = cnt[1]; counts[3] = 0; }
}

extern "C" int mp_obb(const float* verts, float inflate, float* obb, void* stream) {
    hipLaunchKernelGGL(k_obb, dim3(1), dim3(256), 0, (hipStream_t)stream, verts, inflate, obb);
    return (int)hipGetLastError();
}

extern "C" int mp_obb_hull(const double* hull_verts, int n_hull_verts, const double* normals, int n_normals, const double* edge_vec,
                           const double* edge_na, const double* edge_nb, int n_edges, float inflate, double* work, float* obb,
                           void* stream) {
    if (n_hull_verts < 4 || n_normals < 1 || n_edges < 1) return -1;
    const int lds = n_hull_verts * 3 * (int)sizeof(double);
    if (lds > 96 * 1024) return -2;                   // 4096 hull vertices; a posed SMPL body has a few hundred
    hipStream_t st = (hipStream_t)stream;
    MP_LDS_ATTR((k_obb_hull_search), 96 * 1024);
    hipLaunchKernelGGL(k_obb_hull_search, dim3(n_normals), dim3(OBB_T), lds, st, hull_verts, n_hull_verts, normals, edge_vec, edge_na,
                       edge_nb, n_edges, work, (const int*)nullptr, 0LL);
    hipLaunchKernelGGL(k_obb_hull_pick, dim3(1), dim3(OBB_T), 0, st, hull_verts, n_hull_verts, normals, n_normals, edge_vec, work,
                       inflate, obb, (const int*)nullptr, 0LL);
    return (int)hipGetLastError();
}

extern "C" int mp_ray_setup(const float* uv, const float* intrinsics, const float* pose, int n_rays, float radius,
                            float* dirs, float* far, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(k_ray_setup, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, uv, intrinsics, pose,
                       n_rays, radius, dirs, far);
    return (int)hipGetLastError();
}

static int ray_cull(const float* dirs, const float* pose, const float* obb, const float* cbound, const float* far,
                    const float* beta, float near_, int n_rays, int group_size, int* hit_index, int* hit_count, int* inv_index,
                    int* scan_tmp, void* stream) {
    if (n_rays <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int* flag = scan_tmp;                 // [n_rays]
    int* bsum = scan_tmp + n_rays;        // [nb]
    const int nb = (n_rays + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (group_size <= 0) group_size = n_rays;
    hipLaunchKernelGGL(k_ray_box, dim3((n_rays + 255) / 256), dim3(256), 0, st, dirs, pose, obb, n_rays, flag);
    if (cbound)
        hipLaunchKernelGGL(k_ray_near_body, dim3((n_rays + 255) / 256), dim3(256), 0, st, dirs, pose, cbound, far, beta, near_,
                           n_rays, flag);
    hipLaunchKernelGGL(k_group_fallback, dim3((n_rays + group_size - 1) / group_size), dim3(256), 0, st, flag, n_rays,
                       group_size);
    hipLaunchKernelGGL(k_scan_blocks, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, n_rays, bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(64), 0, st, bsum, nb, hit_count);
    hipLaunchKernelGGL(k_scan_scatter, dim3(nb), dim3(SCAN_BLOCK), 0, st, flag, n_rays, bsum, hit_index, inv_index);
    return (int)hipGetLastError();
}

extern "C" int mp_ray_cull(const float* dirs, const float* pose, const float* obb, int n_rays, int group_size,
                           int* hit_index, int* hit_count, int* inv_index, int* scan_tmp, void* stream) {
    return ray_cull(dirs, pose, obb, nullptr, nullptr, nullptr, 0.0f, n_rays, group_size, hit_index, hit_count, inv_index,
                    scan_tmp, stream);
}

extern "C" int mp_ray_cull_near(const float* dirs, const float* pose, const float* obb, const float* cbound, const float* far,
                                const float* beta, float near_, int n_rays, int group_size, int* hit_index, int* hit_count,
                                int* inv_index, int* scan_tmp, void* stream) {
    if (!cbound || !far || !beta) return -1;
    return ray_cull(dirs, pose, obb, cbound, far, beta, near_, n_rays, group_size, hit_index, hit_count, inv_index, scan_tmp,
                    stream);
}

extern "C" int mp_ray_hits_from_index(const int* hit_index, int n_hit, int n_rays, int* hit_count, int* inv_index,
                                      void* stream) {
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hits_from_index, dim3((n_rays + 255) / 256), dim3(256), 0, st, hit_index, n_hit, n_rays,
                       hit_count, inv_index, 0);
    if (n_hit > 0)
        hipLaunchKernelGGL(k_hits_from_index, dim3((n_hit + 255) / 256), dim3(256), 0, st, hit_index, n_hit, n_rays,
                           hit_count, inv_index, 1);
    return (int)hipGetLastError();
}

extern "C" int mp_blend_table(const float* skin_w, const float* tfs, int n_verts, float* table, void* stream) {
    if (n_verts <= 0) return 0;
    hipLaunchKernelGGL(k_blend_table, dim3((n_verts + 255) / 256), dim3(256), 0, (hipStream_t)stream, skin_w, tfs, n_verts,
                       (float4*)table);
    return (int)hipGetLastError();
}

// bin_work (mp_warp_bin_work_bytes(max_rays * n_s) bytes, 16-byte aligned): [BIN_CNT] bin counts, [n] bin << 22 | rank, [n] float4
constexpr int BIN_CNT = (NCC + 127) / 128 * 128;      // bin counters at the head of the work buffer (a multiple of 512 bytes)
static_assert(NC <= 511 && (CL & (CL - 1)) == 0 && CL <= 64 && NC * CL >= V, "cluster layout (include/multiply_hip.h)");
extern "C" int mp_warp_bin_work_bytes(int n_points) { return 4 * BIN_CNT + 4 * ((n_points + 3) / 4 * 4) + 16 * n_points; }
static void warp_bin(const float* dirs, const float* pose, const int* hit_index, const int* hit_count, const float* z, int z_stride,
                     int n_s, int max_rays, const float* cbound, const int* ray_active, const int* launch_active, void* bin_work,
                     hipStream_t st, const float4*& binned, const int*& bincount) {
    const int n = max_rays * n_s;
    int* cnt = (int*)bin_work;
    int* binrank = cnt + BIN_CNT;
    float4* out = (float4*)((char*)bin_work + 4 * BIN_CNT + 4 * ((n + 3) / 4 * 4));
    hipMemsetAsync(cnt, 0, 4 * BIN_CNT, st);
    hipLaunchKernelGGL(k_warp_bin, dim3((n + 1023) / 1024), dim3(1024), 0, st, dirs, pose, hit_index, hit_count, z, z_stride, n_s, max_rays,
                       cbound, ray_active, launch_active, binrank, cnt);
    hipLaunchKernelGGL(k_warp_binned, dim3((n + 255) / 256), dim3(256), 0, st, dirs, pose, hit_index, hit_count, z, z_stride, n_s,
                       max_rays, launch_active, (const int*)binrank, (const int*)cnt, out);
    binned = out;
    bincount = cnt;
}

extern "C" int mp_warp_inverse(const float* pts, const float* dirs, const float* pose, const int* hit_index,
                               const int* hit_count, const float* z, int z_stride, int n_s, int max_rays,
                               const float* vsorted, const float* cbound, const float* blend_table,
                               int mode, const int* ray_active, const int* launch_active, float* xc,
                               unsigned char* outlier, float* sdf_out, int* worklist, int* work_count, void* bin_work,
                               void* stream) {
    // when pts != NULL, max_rays carries the number of explicit points and sdf_out may carry beta for mode 2 (unused)
    if (max_rays <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    MP_LDS_ATTR((k_warp_inverse), WARP_INV_LDS);
    const int n_slab = pts ? (max_rays + 63) / 64 : ray_slabs(max_rays, n_s, mode & 3);
    const int threads = warp_threads(n_slab), nw = threads / 64;
    const float4* binned = nullptr;
    const int* bincount = nullptr;
    if (bin_work && !pts && (mode & 3) == 0) warp_bin(dirs, pose, hit_index, hit_count, z, z_stride, n_s, max_rays, cbound, ray_active,
                                                      launch_active, bin_work, st, binned, bincount);
    hipLaunchKernelGGL(k_warp_inverse, dim3(warp_grid(n_slab, nw)), dim3(threads), WARP_INV_LDS, st, pts, dirs, pose,
                       hit_index, hit_count, z, z_stride, n_s, max_rays, pts ? max_rays : 0, vsorted, cbound,
                       (const float4*)blend_table, mode & 3, ray_active, (const float*)nullptr, launch_active, xc, outlier, (unsigned char*)nullptr, sdf_out,
                       worklist, work_count, (int*)nullptr, binned, bincount);
    return (int)hipGetLastError();
}

// eval-shading variant (mode 2) needs beta; exported separately to keep mp_warp_inverse's signature small
extern "C" int mp_warp_inverse_shade(const float* dirs, const float* pose, const int* hit_index, const int* hit_count,
                                     const float* z, int z_stride, int n_s, int max_rays, const float* vsorted,
                                     const float* cbound, const float* blend_table, int eval_mode,
                                     const float* beta, float* xc, unsigned char* outlier, unsigned char* need_flag,
                                     float* sdf_out, int* worklist, int* work_count, int* nn_index, void* bin_work,
                                     void* stream) {
    if (max_rays <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    MP_LDS_ATTR((k_warp_inverse), WARP_INV_LDS);
    const int n_slab = ray_slabs(max_rays, n_s, eval_mode ? 2 : 0);
    const int threads = warp_threads(n_slab), nw = threads / 64;
    const float4* binned = nullptr;
    const int* bincount = nullptr;
    if (bin_work && !eval_mode) warp_bin(dirs, pose, hit_index, hit_count, z, z_stride, n_s, max_rays, cbound, nullptr, nullptr, bin_work,
                                         st, binned, bincount);
    hipLaunchKernelGGL(k_warp_inverse, dim3(warp_grid(n_slab, nw)), dim3(threads), WARP_INV_LDS, st,
                       (const float*)nullptr, dirs, pose, hit_index, hit_count, z, z_stride, n_s, max_rays, 0, vsorted,
                       cbound, (const float4*)blend_table, eval_mode ? 2 : 0, (const int*)nullptr, beta, (const int*)nullptr, xc, outlier,
                       need_flag, sdf_out, worklist, work_count, nn_index, binned, bincount);
    return (int)hipGetLastError();
}

extern "C" int mp_warp_jacobian(const float* xc, const unsigned char* need, const int* hit_count, int max_rays, int n_s,
                                int n_pts, const float* vsorted_c, const float* cbound_c, const float* blend_table,
                                float* jinv, int* nn_index, const int* seed, const float* verts_c,
                                void* stream) {
    if ((n_s > 0 ? max_rays : n_pts) <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    MP_LDS_ATTR((k_warp_jacobian), WARP_LDS);
    const int n_slab = n_s > 0 ? ray_slabs(max_rays, n_s, 2) : (n_pts + 63) / 64;
    const int threads = warp_threads(n_slab), nw = threads / 64;
    hipLaunchKernelGGL(k_warp_jacobian, dim3(warp_grid(n_slab, nw)), dim3(threads), WARP_LDS, st, xc, need, hit_count,
                       max_rays, n_s, n_pts, vsorted_c, cbound_c, (const float4*)blend_table, jinv, nn_index, seed, verts_c);
    return (int)hipGetLastError();
}

// work (bytes, 8-byte aligned): the workgroups' exchange area (HW_XCH_BYTES; the launcher zeroes its head); then fp64 arrays hv [HW_MAXV][3], normals [HW_MAXF][3], evec / ena / enb
// [3 HW_MAXF / 2][3] each, search scratch [2 HW_MAXF]
constexpr int HW_XCH_BYTES = 256 + 4 * (4 * HW_FRONT + HW_FRONT);
// test hook: every following mp_obb_hull_device call starts with its bodies' "abandoned" words set, i.e. takes the give-up path
// of a wrap whose workgroups never became co-resident (status[3] = 1, obb untouched) without having to starve the GPU for it
static int g_hw_force_abandon = 0;
extern "C" int mp_debug_hull_abandon(int on) { const int was = g_hw_force_abandon; g_hw_force_abandon = on; return was; }

extern "C" int mp_obb_hull_device_work_bytes(void) { return HW_XCH_BYTES + 8 * (3 * HW_MAXV + 3 * HW_MAXF + 3 * (9 * HW_MAXF / 2) + 2 * HW_MAXF); }
extern "C" int mp_obb_hull_device(const float* verts, int n_verts, int n_bodies, float inflate, void* work, float* obb, int* status,
                                  void* stream) {
    if (n_verts < 4 || n_verts >= HW_MAXV || n_verts > 65535 || n_bodies < 1 || n_bodies > 64) return -1;
    hipStream_t st = (hipStream_t)stream;
    int* counts = status;                                   // per body {H, F, E, status, ...}: the caller reads [3] with its other counts
    const long long stride = mp_obb_hull_device_work_bytes() / 8;       // per body, in doubles
    unsigned* xch = (unsigned*)work;                        // barrier counter + front size + failed, records, pivots
    for (int b = 0; b < n_bodies; ++b) {
        hipMemsetAsync((char*)work + 8 * stride * b, 0, 256, st);
        if (g_hw_force_abandon) hipMemsetD32Async((hipDeviceptr_t)((char*)work + 8 * stride * b + 12), 1, 1, st);   // xch[3]
    }
    double* hv = (double*)((char*)work + HW_XCH_BYTES);
    double* normals = hv + 3 * HW_MAXV;
    double* evec = normals + 3 * HW_MAXF;
    double* ena = evec + 9 * HW_MAXF / 2;
    double* enb = ena + 9 * HW_MAXF / 2;
    double* swork = enb + 9 * HW_MAXF / 2;
    MP_LDS_ATTR(k_hull_wrap, HW_LDS);
    hipLaunchKernelGGL(k_hull_wrap, dim3(8 * HW_G * ((n_bodies + 7) / 8)), dim3(HW_T), HW_LDS, st, verts, n_verts, xch, counts, hv,
                       normals, evec, ena, enb, stride, n_bodies);
    MP_LDS_ATTR((k_obb_hull_search), 96 * 1024);
    // the search's LDS tile holds the hull vertices: a closed triangulated surface of F facets has F / 2 + 2 of them
    const int lds = (HW_MAXF / 2 + 2) * 3 * (int)sizeof(double);
    hipLaunchKernelGGL(k_obb_hull_search, dim3(HW_MAXF, n_bodies), dim3(OBB_T), lds, st, hv, 0, normals, evec, ena, enb, 0, swork,
                       (const int*)counts, stride);
    hipLaunchKernelGGL(k_obb_hull_pick, dim3(n_bodies), dim3(OBB_T), 0, st, hv, 0, normals, 0, evec, swork, inflate, obb,
                       (const int*)counts, stride);
    return (int)hipGetLastError();
}
