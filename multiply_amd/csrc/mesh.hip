// In/off-surface ray flags against the current canonical mesh (multiply.py:153-167, training epochs < 250):
//   signed distance of every canonical sample point to a triangle mesh, min over the samples of a ray,
//   off = min > threshold, in = min <= 0.
// The reference calls kaolin 0.13 (third party, not vendored): point_to_mesh_distance = squared distance to the
// closest triangle; check_sign = ray-casting parity (odd number of crossings = inside).  Restated here as:
//   distance : exact closest point on each triangle (region classification after Ericson, Real-Time Collision
//              Detection 5.1.5), minimum over all triangles;
//   sign     : crossings of the ray p + t (1,0,0), t > 0, counted with the half-open rule on the (y,z) projection.
// Brute force over the F triangles (13 776 for the SMPL surface, ~10^5 for a MISE mesh): a workgroup of 256 points
// walks the triangle list through LDS tiles (36 B per triangle, read once per workgroup, broadcast to all lanes).
#include <hip/hip_runtime.h>
#include <float.h>
#include "../../include/multiply_hip.h"

namespace {

constexpr int TB = 256, TILE = 512;

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// squared distance from p to triangle (a, b, c)
__device__ __forceinline__ float tri_dist2(const float* p, const float* a, const float* b, const float* c) {
    float ab[3], ac[3], ap[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; }
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    float q[3];
    if (d1 <= 0.f && d2 <= 0.f) { for (int i = 0; i < 3; ++i) q[i] = a[i]; }
    else {
        float bp[3];
        for (int i = 0; i < 3; ++i) bp[i] = p[i] - b[i];
        const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        if (d3 >= 0.f && d4 <= d3) { for (int i = 0; i < 3; ++i) q[i] = b[i]; }
        else {
            const float vc = d1 * d4 - d3 * d2;
            if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
                const float v = d1 / (d1 - d3);
                for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i];
            } else {
                float cp[3];
                for (int i = 0; i < 3; ++i) cp[i] = p[i] - c[i];
                const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
                if (d6 >= 0.f && d5 <= d6) { for (int i = 0; i < 3; ++i) q[i] = c[i]; }
                else {
                    const float vb = d5 * d2 - d1 * d6;
                    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
                        const float w = d2 / (d2 - d6);
                        for (int i = 0; i < 3; ++i) q[i] = a[i] + w * ac[i];
                    } else {
                        const float va = d3 * d6 - d5 * d4;
                        if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
                            const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
                            for (int i = 0; i < 3; ++i) q[i] = b[i] + w * (c[i] - b[i]);
                        } else {
                            const float den = 1.0f / (va + vb + vc);
                            const float v = vb * den, w = vc * den;
                            for (int i = 0; i < 3; ++i) q[i] = a[i] + ab[i] * v + ac[i] * w;
                        }
                    }
                }
            }
        }
    }
    const float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
    return dx * dx + dy * dy + dz * dz;
}

// does the ray p + t (1,0,0), t > 0 cross the triangle?  (y,z) projection, half-open edges: an edge (u,v) counts when
// exactly one endpoint has y > p.y; the crossing is inside when the edge functions agree in sign.
__device__ __forceinline__ bool ray_x_crosses(const float* p, const float* a, const float* b, const float* c) {
    // 2D point-in-triangle in (y,z) by the crossing-number rule along +z, then the x of the plane point
    int cn = 0;
    const float* v[3] = {a, b, c};
    for (int e = 0; e < 3; ++e) {
        const float* u = v[e];
        const float* w = v[(e + 1) % 3];
        const bool uy = u[1] > p[1], wy = w[1] > p[1];
        if (uy != wy) {
            const float t = (p[1] - u[1]) / (w[1] - u[1]);
            const float zc = u[2] + t * (w[2] - u[2]);
            if (zc > p[2]) ++cn;
        }
    }
    if ((cn & 1) == 0) return false;
    // plane: n . (x - a) = 0 -> x at (p.y, p.z)
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
    if (nx == 0.f) return false;   // triangle parallel to the ray
    const float x = a[0] - (ny * (p[1] - a[1]) + nz * (p[2] - a[2])) / nx;
    return x > p[0];
}

__global__ __launch_bounds__(TB) void k_mesh_sdist(const float* __restrict__ pts, int n, const float* __restrict__ fv, int F,
                                                   float* __restrict__ sdist) {
    __shared__ float tri[TILE * 9];
    const int i = blockIdx.x * TB + threadIdx.x;
    float p[3] = {0.f, 0.f, 0.f};
    if (i < n) { p[0] = pts[3 * (size_t)i]; p[1] = pts[3 * (size_t)i + 1]; p[2] = pts[3 * (size_t)i + 2]; }
    float best = FLT_MAX;
    int crossings = 0;
    for (int f0 = 0; f0 < F; f0 += TILE) {
        const int nt = min(TILE, F - f0);
        __syncthreads();
        for (int k = threadIdx.x; k < nt * 9; k += TB) tri[k] = fv[(size_t)f0 * 9 + k];
        __syncthreads();
        if (i < n) {
            for (int t = 0; t < nt; ++t) {
                const float* a = tri + 9 * t;
                best = fminf(best, tri_dist2(p, a, a + 3, a + 6));
                crossings += ray_x_crosses(p, a, a + 3, a + 6) ? 1 : 0;
            }
        }
    }
    if (i < n) sdist[i] = (crossings & 1) ? -sqrtf(best) : sqrtf(best);
}

__global__ void k_ray_flags(const float* __restrict__ sdist, int n_rays, int n_s, float threshold,
                            unsigned char* __restrict__ off, unsigned char* __restrict__ in) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_rays) return;
    float m = FLT_MAX;
    for (int s = 0; s < n_s; ++s) m = fminf(m, sdist[(size_t)k * n_s + s]);
    off[k] = m > threshold ? 1 : 0;
    in[k] = m <= 0.0f ? 1 : 0;
}

}  // namespace

extern "C" int mp_mesh_signed_distance(const float* pts, int n, const float* face_verts, int n_faces, float* sdist,
                                       void* stream) {
    if (n <= 0) return 0;
    if (n_faces <= 0) return -1;
    hipLaunchKernelGGL(k_mesh_sdist, dim3((n + TB - 1) / TB), dim3(TB), 0, (hipStream_t)stream, pts, n, face_verts, n_faces,
                       sdist);
    return (int)hipGetLastError();
}

extern "C" int mp_mesh_ray_flags(const float* sdist, int n_rays, int n_s, float threshold, unsigned char* off,
                                 unsigned char* in, void* stream) {
    if (n_rays <= 0) return 0;
    hipLaunchKernelGGL(k_ray_flags, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, sdist, n_rays, n_s,
                       threshold, off, in);
    return (int)hipGetLastError();
}

// ---- general skinning-weight query (deformer.py:37-50) with K <= 8 nearest vertices, and skinning with explicit weights
// (deformer.py:72-88).  Not on the render/training path (K = 1 there, fused into the warp kernels); the reference's
// trainer switches K to 7 when it transfers weights to an extracted mesh (multiply_model.py:1174-1177).
namespace {
constexpr int QK_MAX = 8, QV_TILE = 2048;

__global__ __launch_bounds__(256) void k_query_weights(const float* __restrict__ pts, int n, const float* __restrict__ verts,
                                                       int V, const float* __restrict__ skin_w, int K,
                                                       float* __restrict__ weights, unsigned char* __restrict__ outlier) {
    __shared__ float vt[QV_TILE * 3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float p[3] = {0.f, 0.f, 0.f};
    if (i < n) { p[0] = pts[3 * (size_t)i]; p[1] = pts[3 * (size_t)i + 1]; p[2] = pts[3 * (size_t)i + 2]; }
    float bd[QK_MAX];
    int bi[QK_MAX];
    for (int k = 0; k < QK_MAX; ++k) { bd[k] = FLT_MAX; bi[k] = -1; }
    for (int v0 = 0; v0 < V; v0 += QV_TILE) {
        const int nv = min(QV_TILE, V - v0);
        __syncthreads();
        for (int k = threadIdx.x; k < nv * 3; k += blockDim.x) vt[k] = verts[(size_t)v0 * 3 + k];
        __syncthreads();
        if (i < n)
            for (int v = 0; v < nv; ++v) {
                const float dx = p[0] - vt[3 * v], dy = p[1] - vt[3 * v + 1], dz = p[2] - vt[3 * v + 2];
                float d = dx * dx + dy * dy + dz * dz;
                if (d < bd[K - 1]) {   // insertion into the sorted list (ties keep the lower vertex id first)
                    int id = v0 + v;
#pragma unroll
                    for (int k = 0; k < QK_MAX; ++k)
                        if (k < K && d < bd[k]) {
                            const float td = bd[k]; const int ti = bi[k];
                            bd[k] = d; bi[k] = id; d = td; id = ti;
                        }
                }
            }
    }
    if (i >= n) return;
    float conf[QK_MAX], csum = 0.f;
    for (int k = 0; k < K; ++k) { conf[k] = expf(-fminf(bd[k], 4.0f)); csum += conf[k]; }
    for (int j = 0; j < 24; ++j) {
        float w = 0.f;
        for (int k = 0; k < K; ++k) w += skin_w[(size_t)bi[k] * 24 + j] * (conf[k] / csum);
        weights[(size_t)i * 24 + j] = w;
    }
    if (outlier) outlier[i] = sqrtf(fminf(bd[0], 4.0f)) > 0.1f ? 1 : 0;
}

__global__ void k_skinning(const float* __restrict__ pts, const float* __restrict__ weights, int n,
                           const float* __restrict__ tfs, int inverse, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float T[16];
    for (int e = 0; e < 16; ++e) T[e] = 0.f;
    for (int j = 0; j < 24; ++j) {
        const float w = weights[(size_t)i * 24 + j];
        if (w != 0.f) for (int e = 0; e < 16; ++e) T[e] += w * tfs[16 * j + e];
    }
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    float o[3];
    if (!inverse) {
        for (int a = 0; a < 3; ++a) o[a] = T[4 * a] * x + T[4 * a + 1] * y + T[4 * a + 2] * z + T[4 * a + 3];
    } else {   // first three components of T^-1 [x,1] for T = [R t; 0 0 0 s]: R^-1 (x - t / s)
        const float a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], k = T[10];
        const float c0 = e * k - f * h, c1 = f * g - d * k, c2 = d * h - e * g;
        const float r = 1.0f / (a * c0 + b * c1 + c * c2);
        const float qx = x - T[3] / T[15], qy = y - T[7] / T[15], qz = z - T[11] / T[15];
        o[0] = (c0 * qx + (c * h - b * k) * qy + (b * f - c * e) * qz) * r;
        o[1] = (c1 * qx + (a * k - c * g) * qy + (c * d - a * f) * qz) * r;
        o[2] = (c2 * qx + (b * g - a * h) * qy + (a * e - b * d) * qz) * r;
    }
    for (int a2 = 0; a2 < 3; ++a2) out[3 * (size_t)i + a2] = o[a2];
}
}  // namespace

extern "C" int mp_query_weights(const float* pts, int n, const float* verts, int n_verts, const float* skin_w, int K,
                                float* weights, unsigned char* outlier, void* stream) {
    if (n <= 0) return 0;
    if (K < 1 || K > QK_MAX) return -1;
    hipLaunchKernelGGL(k_query_weights, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, n, verts, n_verts,
                       skin_w, K, weights, outlier);
    return (int)hipGetLastError();
}

extern "C" int mp_skinning(const float* pts, const float* weights, int n, const float* tfs, int inverse, float* out,
                           void* stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_skinning, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, weights, n, tfs, inverse,
                       out);
    return (int)hipGetLastError();
}
