// Per-person mesh z-buffer (the hard rasterisation the reference asks of pytorch3d: code/lib/model/render.py:64-66,
// 134-157 `render_multiple_depth_map`, callers multiply_model.py:396, :634, :875).
//
// pytorch3d is a third-party dependency that is not under /root/reference (unpinned, README.md:15): this file restates
// its published rasterisation rule for blur_radius = 0, perspective_correct = True, cull_backfaces = False:
//   * a pixel (row i, column j) is sampled at its centre (j + 0.5, i + 0.5) in screen space;
//   * face f covers the pixel when the three barycentric coordinates  w_k = edge_k(p) / (area + 1e-8)  are > 0
//     (either winding), faces with |area| <= 1e-8 are skipped;
//   * depth = 1 / sum_k (w_k / z_k)  written as pytorch3d does (w_k * z_l * z_m over their sum, clamped at 1e-8),
//     faces behind the image plane (depth < 0) are skipped; the nearest face wins; empty pixels hold -1.
// The barycentric rule is invariant to the affine pixel <-> NDC map, so the kernel works in pixel units.
//
// Layout for MI355X: the work is one gather of 3 vertices per face plus a handful of pixel tests (a body mesh at image
// scale has ~1 pixel per face), so it is scattered rather than tiled: one thread per face walks the face's pixel box and
// resolves visibility with a 64-bit atomicMin on (depth bits << 32 | face id) -- positive floats order like their bit
// patterns, and ties go to the lower face id, so the result does not depend on scheduling.  Faces with a large box are
// queued and swept by a whole workgroup each in a second launch.  A final pass unpacks depth / face id and recomputes the
// winner's perspective-correct barycentrics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/multiply_hip.h"
#include "common.hpp"

namespace {

constexpr int SMALL_BOX = 48;   // pixel tests a single thread does before the face goes to the workgroup queue
constexpr float K_EPS = 1e-8f;

struct Cam {
    float R[9], T[3], fx, fy, cx, cy, z_clip;
};

struct Tri {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2, area;
};

__device__ __forceinline__ void project(const Cam& c, const float* __restrict__ v, float& x, float& y, float& z) {
    const float X = c.R[0] * v[0] + c.R[1] * v[1] + c.R[2] * v[2] + c.T[0];
    const float Y = c.R[3] * v[0] + c.R[4] * v[1] + c.R[5] * v[2] + c.T[1];
    z = c.R[6] * v[0] + c.R[7] * v[1] + c.R[8] * v[2] + c.T[2];
    x = c.fx * X / z + c.cx;
    y = c.fy * Y / z + c.cy;
}

// edge(p; a, b) of pytorch3d's EdgeFunctionForward
__device__ __forceinline__ float edge(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ bool load_tri(const Cam& c, const float* __restrict__ verts, const int* __restrict__ faces, int f,
                                         Tri& t) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    project(c, verts + 3 * (size_t)i0, t.x0, t.y0, t.z0);
    project(c, verts + 3 * (size_t)i1, t.x1, t.y1, t.z1);
    project(c, verts + 3 * (size_t)i2, t.x2, t.y2, t.z2);
    if (!(fminf(t.z0, fminf(t.z1, t.z2)) >= c.z_clip)) return false;   // dropped, not clipped (see the header)
    t.area = edge(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1);
    return fabsf(t.area) > K_EPS;
}

// perspective-correct barycentrics and depth of pixel centre (px, py); false = not covered
__device__ __forceinline__ bool cover(const Tri& t, float px, float py, float& b0, float& b1, float& b2, float& pz) {
    const float a = t.area + K_EPS;
    const float w0 = edge(px, py, t.x1, t.y1, t.x2, t.y2) / a;
    const float w1 = edge(px, py, t.x2, t.y2, t.x0, t.y0) / a;
    const float w2 = edge(px, py, t.x0, t.y0, t.x1, t.y1) / a;
    if (!(w0 > 0.f && w1 > 0.f && w2 > 0.f)) return false;
    const float t0 = w0 * t.z1 * t.z2, t1 = t.z0 * w1 * t.z2, t2 = t.z0 * t.z1 * w2;
    const float d = fmaxf(t0 + t1 + t2, K_EPS);
    b0 = t0 / d; b1 = t1 / d; b2 = t2 / d;
    pz = b0 * t.z0 + b1 * t.z1 + b2 * t.z2;
    return pz >= 0.f;
}

__device__ __forceinline__ bool pixel_box(const Tri& t, int H, int W, int& c0, int& c1, int& r0, int& r1) {
    const float xmin = fminf(t.x0, fminf(t.x1, t.x2)), xmax = fmaxf(t.x0, fmaxf(t.x1, t.x2));
    const float ymin = fminf(t.y0, fminf(t.y1, t.y2)), ymax = fmaxf(t.y0, fmaxf(t.y1, t.y2));
    if (!(xmax >= 0.f && ymax >= 0.f && xmin <= (float)W && ymin <= (float)H)) return false;   // also rejects NaN
    c0 = max(0, (int)ceilf(xmin - 0.5f));      // pixel centres c + 0.5 inside [xmin, xmax]
    c1 = min(W - 1, (int)floorf(xmax - 0.5f));
    r0 = max(0, (int)ceilf(ymin - 0.5f));
    r1 = min(H - 1, (int)floorf(ymax - 0.5f));
    return c0 <= c1 && r0 <= r1;
}

__device__ __forceinline__ void splat(const Tri& t, int f, int r, int c, int W, unsigned long long* __restrict__ keys) {
    float b0, b1, b2, pz;
    if (cover(t, c + 0.5f, r + 0.5f, b0, b1, b2, pz))
        atomicMin(keys + (size_t)r * W + c, ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)f);
}

__global__ void k_raster_faces(const float* __restrict__ verts, const int* __restrict__ faces, int F, Cam cam, int H, int W,
                               unsigned long long* __restrict__ keys, int* __restrict__ big) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    Tri t;
    int c0, c1, r0, r1;
    if (!load_tri(cam, verts, faces, f, t) || !pixel_box(t, H, W, c0, c1, r0, r1)) return;
    if ((c1 - c0 + 1) * (long long)(r1 - r0 + 1) > SMALL_BOX) {
        big[1 + atomicAdd(big, 1)] = f;
        return;
    }
    for (int r = r0; r <= r1; ++r)
        for (int c = c0; c <= c1; ++c) splat(t, f, r, c, W, keys);
}

// one workgroup per queued face; the queue length lives on the device
__global__ void k_raster_big(const float* __restrict__ verts, const int* __restrict__ faces, Cam cam, int H, int W,
                             unsigned long long* __restrict__ keys, const int* __restrict__ big) {
    const int n = big[0];
    for (int q = blockIdx.x; q < n; q += gridDim.x) {
        const int f = big[1 + q];
        Tri t;
        int c0, c1, r0, r1;
        if (!load_tri(cam, verts, faces, f, t) || !pixel_box(t, H, W, c0, c1, r0, r1)) continue;
        const int bw = c1 - c0 + 1;
        const long long np = (long long)bw * (r1 - r0 + 1);
        for (long long i = threadIdx.x; i < np; i += blockDim.x) splat(t, f, r0 + (int)(i / bw), c0 + (int)(i % bw), W, keys);
    }
}

__global__ void k_raster_resolve(const float* __restrict__ verts, const int* __restrict__ faces, Cam cam, int H, int W,
                                 const unsigned long long* __restrict__ keys, float* __restrict__ zbuf,
                                 int* __restrict__ pix_to_face, float* __restrict__ bary) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (size_t)H * W) return;
    const unsigned long long k = keys[p];
    float z = -1.f, b0 = -1.f, b1 = -1.f, b2 = -1.f;
    int f = -1;
    if (k != ~0ull) {
        f = (int)(unsigned)(k & 0xffffffffull);
        Tri t;
        load_tri(cam, verts, faces, f, t);
        float pz;
        cover(t, (float)(p % W) + 0.5f, (float)(p / W) + 0.5f, b0, b1, b2, pz);
        z = __uint_as_float((unsigned)(k >> 32));
    }
    zbuf[p] = z;
    if (pix_to_face) pix_to_face[p] = f;
    if (bary) { bary[3 * p] = b0; bary[3 * p + 1] = b1; bary[3 * p + 2] = b2; }
}

// ---------------------------------------------------------------------------------------------------------------------
// Soft silhouette render (code/lib/model/render.py:79-105, 121-133 `softrender_multiple_meshes`: pytorch3d's MeshRasterizer
// with blur_radius = log(1/1e-4 - 1) sigma and faces_per_pixel = 100, SoftPhongShader under white ambient light =
// softmax_rgb_blend of the interpolated vertex colours).  Restated like the z-buffer above (pytorch3d is absent):
//   * a face is a candidate of a pixel when the pixel centre lies in the face's box widened by sqrt(blur_radius) and either
//     inside the face or nearer than blur_radius (SQUARED distance, NDC units: the shorter image side spans [-1, 1]) to its
//     outline; its depth is interpolated with the perspective-correct barycentrics clipped to >= 0 and renormalised;
//   * the K candidates nearest in depth are kept (ties: lower face id);
//   * prob_k = sigmoid(-d_k / sigma) (d < 0 inside), alpha = prod (1 - prob_k), weights prob_k exp((zinv_k - zinv_max) / gamma)
//     with zinv = (zfar - z) / (zfar - znear), background weight delta = max(exp((1e-10 - zinv_max) / gamma), 1e-10).
//
// Layout: pixel-centric, because every pixel needs ITS K nearest candidates: faces are binned to 8 x 8-pixel tiles (count,
// scan, fill: three small launches, list order arbitrary), one 64-lane workgroup per tile stages 64 faces at a time in LDS
// (each lane projects one) and every lane keeps its pixel's K best (depth, distance, face) in LDS columns -- K x 64 x 12 B =
// 75 KiB for K = 100 -- replacing the current worst when full.  The selected SET does not depend on the list order; the
// blend accumulates in double so that the order of the sums does not show in the float result.
constexpr int SOFT_TILE = 8, SOFT_MAX_K = 100;

struct SoftTri {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2, area;
    int ok;
};

__device__ __forceinline__ void load_soft_tri(const Cam& c, const float* __restrict__ verts, const int* __restrict__ faces, int f,
                                              float sc, SoftTri& t) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    project(c, verts + 3 * (size_t)i0, t.x0, t.y0, t.z0);
    project(c, verts + 3 * (size_t)i1, t.x1, t.y1, t.z1);
    project(c, verts + 3 * (size_t)i2, t.x2, t.y2, t.z2);
    t.x0 *= sc; t.y0 *= sc; t.x1 *= sc; t.y1 *= sc; t.x2 *= sc; t.y2 *= sc;     // pixel units -> NDC units
    t.area = edge(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1);
    t.ok = (fminf(t.z0, fminf(t.z1, t.z2)) >= c.z_clip) && fabsf(t.area) > K_EPS;
}

// tiles whose pixels can pass the widened-box test (one pixel of slack on every side against rounding)
__device__ __forceinline__ bool soft_tile_range(const SoftTri& t, float sc, float rad, int H, int W, int& tc0, int& tc1, int& tr0,
                                                int& tr1) {
    const float xmin = fminf(t.x0, fminf(t.x1, t.x2)) - rad, xmax = fmaxf(t.x0, fmaxf(t.x1, t.x2)) + rad;
    const float ymin = fminf(t.y0, fminf(t.y1, t.y2)) - rad, ymax = fmaxf(t.y0, fmaxf(t.y1, t.y2)) + rad;
    if (!(xmax >= 0.f && ymax >= 0.f && xmin <= W * sc && ymin <= H * sc)) return false;      // also rejects NaN
    const float lim = 1.0e9f;
    const int c0 = max(0, (int)floorf(fminf(fmaxf(xmin / sc - 0.5f, -lim), lim)) - 1);
    const int c1 = min(W - 1, (int)ceilf(fminf(fmaxf(xmax / sc - 0.5f, -lim), lim)) + 1);
    const int r0 = max(0, (int)floorf(fminf(fmaxf(ymin / sc - 0.5f, -lim), lim)) - 1);
    const int r1 = min(H - 1, (int)ceilf(fminf(fmaxf(ymax / sc - 0.5f, -lim), lim)) + 1);
    if (c0 > c1 || r0 > r1) return false;
    tc0 = c0 / SOFT_TILE; tc1 = c1 / SOFT_TILE; tr0 = r0 / SOFT_TILE; tr1 = r1 / SOFT_TILE;
    return true;
}

// FILL = false: tile_n[tile] += 1 per overlapped tile;  FILL = true: list[offsets[tile] + tile_n[tile]++] = face
template <bool FILL>
__global__ void k_soft_bin(const float* __restrict__ verts, const int* __restrict__ faces, int F, Cam cam, int H, int W, float sc,
                           float rad, int tiles_x, int* __restrict__ tile_n, const int* __restrict__ offsets, int* __restrict__ list) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    SoftTri t;
    load_soft_tri(cam, verts, faces, f, sc, t);
    int tc0, tc1, tr0, tr1;
    if (!t.ok || !soft_tile_range(t, sc, rad, H, W, tc0, tc1, tr0, tr1)) return;
    for (int tr = tr0; tr <= tr1; ++tr)
        for (int tc = tc0; tc <= tc1; ++tc) {
            const int tile = tr * tiles_x + tc;
            const int k = atomicAdd(tile_n + tile, 1);
            if (FILL) list[offsets[tile] + k] = f;
        }
}

// exclusive scan of tile_n[0..T) into offsets[0..T], offsets[T] = total; tile_n is cleared for the fill pass.  One workgroup.
__global__ __launch_bounds__(256) void k_soft_scan(int* __restrict__ tile_n, int T, int* __restrict__ offsets) {
    __shared__ long long part[256];
    const int t = threadIdx.x, per = (T + 255) / 256, b = t * per, e = min(T, b + per);
    long long s = 0;
    for (int i = b; i < e; ++i) s += tile_n[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) { const long long v = part[i]; part[i] = run; run += v; }
        offsets[T] = run > 0x7fffffffLL ? -1 : (int)run;          // -1: more list entries than an int holds
    }
    __syncthreads();
    long long run = part[t];
    for (int i = b; i < e; ++i) {
        offsets[i] = (int)(run > 0x7fffffffLL ? 0x7fffffffLL : run);
        run += tile_n[i];
        tile_n[i] = 0;
    }
}

struct SoftParams {
    float sc, blur, sigma, gamma, znear, zfar, bg[3];
    int K;
};

// candidate test of pytorch3d's CheckPixelInsideFace; -> keep, with depth pz, signed squared distance sd and (optionally) the
// clipped barycentrics
__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay, l2 = bax * bax + bay * bay;
    if (l2 <= K_EPS) return (px - bx) * (px - bx) + (py - by) * (py - by);
    const float tt = fminf(fmaxf(((px - ax) * bax + (py - ay) * bay) / l2, 0.f), 1.f);
    const float qx = ax + tt * bax, qy = ay + tt * bay;
    return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}
__device__ __forceinline__ bool soft_candidate(const SoftTri& t, float px, float py, float rad, float blur, float& pz, float& sd,
                                               float& b0, float& b1, float& b2) {
    if (!t.ok) return false;
    if (!(px >= fminf(t.x0, fminf(t.x1, t.x2)) - rad && px <= fmaxf(t.x0, fmaxf(t.x1, t.x2)) + rad &&
          py >= fminf(t.y0, fminf(t.y1, t.y2)) - rad && py <= fmaxf(t.y0, fmaxf(t.y1, t.y2)) + rad))
        return false;
    const float a = t.area + K_EPS;
    const float w0 = edge(px, py, t.x1, t.y1, t.x2, t.y2) / a;
    const float w1 = edge(px, py, t.x2, t.y2, t.x0, t.y0) / a;
    const float w2 = edge(px, py, t.x0, t.y0, t.x1, t.y1) / a;
    const bool inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
    const float dist = fminf(fminf(seg_dist2(px, py, t.x0, t.y0, t.x1, t.y1), seg_dist2(px, py, t.x0, t.y0, t.x2, t.y2)),
                             seg_dist2(px, py, t.x1, t.y1, t.x2, t.y2));
    if (!inside && !(dist < blur)) return false;
    const float t0 = w0 * t.z1 * t.z2, t1 = t.z0 * w1 * t.z2, t2 = t.z0 * t.z1 * w2;
    const float d = fmaxf(t0 + t1 + t2, K_EPS);
    b0 = fmaxf(t0 / d, 0.f); b1 = fmaxf(t1 / d, 0.f); b2 = fmaxf(t2 / d, 0.f);
    const float bs = fmaxf(b0 + b1 + b2, 1e-5f);
    b0 /= bs; b1 /= bs; b2 /= bs;
    pz = b0 * t.z0 + b1 * t.z1 + b2 * t.z2;
    if (!(pz >= 0.f)) return false;
    sd = inside ? -dist : dist;
    return true;
}

__global__ __launch_bounds__(64) void k_soft_blend(const float* __restrict__ verts, const int* __restrict__ faces,
                                                   const float* __restrict__ colors, Cam cam, int H, int W, SoftParams sp, int tiles_x,
                                                   const int* __restrict__ offsets, const int* __restrict__ list,
                                                   float* __restrict__ image, int* __restrict__ sel) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = sp.K, lane = threadIdx.x;
    float* zl = smem;                       // [K][64]
    float* dl = zl + K * 64;                // [K][64]
    int* fl = (int*)(dl + K * 64);          // [K][64]
    SoftTri* stage = (SoftTri*)(fl + K * 64);
    int* stage_f = (int*)(stage + 64);
    const int tile = blockIdx.x, tr = tile / tiles_x, tc = tile % tiles_x;
    const int r = tr * SOFT_TILE + lane / SOFT_TILE, c = tc * SOFT_TILE + lane % SOFT_TILE;
    const bool live = r < H && c < W;
    const float px = (c + 0.5f) * sp.sc, py = (r + 0.5f) * sp.sc, rad = sqrtf(sp.blur);
    int n = 0, imax = 0, fmax = 0;
    float zmax = 0.f;
    const int beg = offsets[tile], end = offsets[tile + 1];
    for (int base = beg; base < end; base += 64) {
        const int m = min(64, end - base);
        __syncthreads();
        if (lane < m) {
            const int f = list[base + lane];
            stage_f[lane] = f;
            load_soft_tri(cam, verts, faces, f, sp.sc, stage[lane]);
        }
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < m; ++j) {
            float pz, sd, b0, b1, b2;
            if (!soft_candidate(stage[j], px, py, rad, sp.blur, pz, sd, b0, b1, b2)) continue;
            const int f = stage_f[j];
            int slot = -1;
            if (n < K) slot = n++;
            else if (pz < zmax || (pz == zmax && f < fmax)) slot = imax;
            if (slot < 0) continue;
            zl[slot * 64 + lane] = pz; dl[slot * 64 + lane] = sd; fl[slot * 64 + lane] = f;
            if (n == K) {                                   // the entry to evict next: largest (depth, face id)
                zmax = zl[lane]; fmax = fl[lane]; imax = 0;
                for (int k = 1; k < K; ++k) {
                    const float zk = zl[k * 64 + lane];
                    const int fk = fl[k * 64 + lane];
                    if (zk > zmax || (zk == zmax && fk > fmax)) { zmax = zk; fmax = fk; imax = k; }
                }
            }
        }
    }
    if (!live) return;
    const float eps = 1e-10f, zr = sp.zfar - sp.znear;
    float zinv_max = n < K ? 0.f : -INFINITY;                // empty slots count as zinv = 0
    for (int k = 0; k < n; ++k) zinv_max = fmaxf(zinv_max, (sp.zfar - zl[k * 64 + lane]) / zr);
    zinv_max = fmaxf(zinv_max, eps);
    double alpha = 1.0, wsum = 0.0, cr = 0.0, cg = 0.0, cb = 0.0;
    for (int k = 0; k < n; ++k) {
        const int f = fl[k * 64 + lane];
        const float prob = 1.f / (1.f + expf(dl[k * 64 + lane] / sp.sigma));
        const float w = prob * expf(((sp.zfar - zl[k * 64 + lane]) / zr - zinv_max) / sp.gamma);
        SoftTri t;
        load_soft_tri(cam, verts, faces, f, sp.sc, t);
        float pz, sd, b0, b1, b2;
        soft_candidate(t, px, py, rad, sp.blur, pz, sd, b0, b1, b2);
        const float* c0 = colors + 3 * (size_t)faces[3 * f];
        const float* c1 = colors + 3 * (size_t)faces[3 * f + 1];
        const float* c2 = colors + 3 * (size_t)faces[3 * f + 2];
        alpha *= (double)(1.f - prob);
        wsum += (double)w;
        cr += (double)(w * (b0 * c0[0] + b1 * c1[0] + b2 * c2[0]));
        cg += (double)(w * (b0 * c0[1] + b1 * c1[1] + b2 * c2[1]));
        cb += (double)(w * (b0 * c0[2] + b1 * c1[2] + b2 * c2[2]));
        if (sel) sel[((size_t)r * W + c) * K + k] = f;
    }
    if (sel)
        for (int k = n; k < K; ++k) sel[((size_t)r * W + c) * K + k] = -1;
    const float delta = fmaxf(expf((eps - zinv_max) / sp.gamma), eps);
    const double den = wsum + (double)delta;
    float* o = image + ((size_t)r * W + c) * 4;
    o[0] = (float)((cr + (double)(delta * sp.bg[0])) / den);
    o[1] = (float)((cg + (double)(delta * sp.bg[1])) / den);
    o[2] = (float)((cb + (double)(delta * sp.bg[2])) / den);
    o[3] = (float)(1.0 - alpha);
}

__host__ Cam cam_from_host(const float* cam_host, float z_clip) {
    Cam cam;
    for (int i = 0; i < 9; ++i) cam.R[i] = cam_host[i];
    for (int i = 0; i < 3; ++i) cam.T[i] = cam_host[9 + i];
    cam.fx = cam_host[12]; cam.fy = cam_host[13]; cam.cx = cam_host[14]; cam.cy = cam_host[15];
    cam.z_clip = z_clip;
    return cam;
}

}  // namespace

extern "C" int mp_raster_zbuf(const float* verts, int n_verts, const int* faces, int n_faces, const float* cam_host,
                              float z_clip, int H, int W, unsigned long long* keys, int* big, float* zbuf, int* pix_to_face,
                              float* bary, void* stream) {
    if (H <= 0 || W <= 0 || n_verts < 0 || n_faces < 0 || !cam_host || !keys || !big || !zbuf) return -1;
    hipStream_t st = (hipStream_t)stream;
    Cam cam;
    for (int i = 0; i < 9; ++i) cam.R[i] = cam_host[i];
    for (int i = 0; i < 3; ++i) cam.T[i] = cam_host[9 + i];
    cam.fx = cam_host[12]; cam.fy = cam_host[13]; cam.cx = cam_host[14]; cam.cy = cam_host[15];
    cam.z_clip = z_clip;
    const size_t npix = (size_t)H * W;
    hipError_t e = hipMemsetAsync(keys, 0xff, npix * sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(big, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (n_faces > 0) {
        hipLaunchKernelGGL(k_raster_faces, dim3((n_faces + 255) / 256), dim3(256), 0, st, verts, faces, n_faces, cam, H, W,
                           keys, big);
        hipLaunchKernelGGL(k_raster_big, dim3(n_faces < 2048 ? n_faces : 2048), dim3(256), 0, st, verts, faces, cam, H, W,
                           keys, big);
    }
    hipLaunchKernelGGL(k_raster_resolve, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, verts, faces, cam, H, W, keys,
                       zbuf, pix_to_face, bary);
    return (int)hipGetLastError();
}

extern "C" int mp_raster_soft_bins(const float* verts, int n_verts, const int* faces, int n_faces, const float* cam_host,
                                   float z_clip, int H, int W, float blur_radius, int* tile_n, int* offsets, void* stream) {
    if (H <= 0 || W <= 0 || n_verts < 0 || n_faces < 0 || !cam_host || !tile_n || !offsets || !(blur_radius >= 0.f)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const Cam cam = cam_from_host(cam_host, z_clip);
    const int tx = (W + SOFT_TILE - 1) / SOFT_TILE, ty = (H + SOFT_TILE - 1) / SOFT_TILE, T = tx * ty;
    const float sc = 2.0f / (float)(H < W ? H : W);
    hipError_t e = hipMemsetAsync(tile_n, 0, (size_t)T * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (n_faces > 0)
        hipLaunchKernelGGL(k_soft_bin<false>, dim3((n_faces + 255) / 256), dim3(256), 0, st, verts, faces, n_faces, cam, H, W, sc,
                           sqrtf(blur_radius), tx, tile_n, (const int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(k_soft_scan, dim3(1), dim3(256), 0, st, tile_n, T, offsets);
    return (int)hipGetLastError();
}

extern "C" int mp_raster_soft(const float* verts, int n_verts, const int* faces, int n_faces, const float* colors,
                              const float* cam_host, float z_clip, int H, int W, float sigma, float gamma, float blur_radius,
                              int faces_per_pixel, float znear, float zfar, const float* background_host, int* tile_n,
                              const int* offsets, int* list, float* image, int* sel, void* stream) {
    if (H <= 0 || W <= 0 || n_verts < 0 || n_faces < 0 || !cam_host || !tile_n || !offsets || !image || !background_host) return -1;
    if (faces_per_pixel < 1 || faces_per_pixel > SOFT_MAX_K || !(sigma > 0.f) || !(gamma > 0.f) || !(blur_radius >= 0.f) ||
        !(zfar > znear))
        return -1;
    if (n_faces > 0 && (!verts || !faces || !colors || !list)) return -1;
    hipStream_t st = (hipStream_t)stream;
    const Cam cam = cam_from_host(cam_host, z_clip);
    const int tx = (W + SOFT_TILE - 1) / SOFT_TILE, ty = (H + SOFT_TILE - 1) / SOFT_TILE, T = tx * ty;
    SoftParams sp;
    sp.sc = 2.0f / (float)(H < W ? H : W);
    sp.blur = blur_radius; sp.sigma = sigma; sp.gamma = gamma; sp.znear = znear; sp.zfar = zfar; sp.K = faces_per_pixel;
    for (int i = 0; i < 3; ++i) sp.bg[i] = background_host[i];
    if (n_faces > 0)
        hipLaunchKernelGGL(k_soft_bin<true>, dim3((n_faces + 255) / 256), dim3(256), 0, st, verts, faces, n_faces, cam, H, W, sp.sc,
                           sqrtf(blur_radius), tx, tile_n, offsets, list);
    const int lds = faces_per_pixel * 64 * 12 + 64 * (int)(sizeof(SoftTri) + sizeof(int));
    // the attribute is set ONCE per device (MP_LDS_ATTR): to the largest size any call may ask for (K = SOFT_MAX_K), not to this
    // call's -- a first call with K = 10 (11 KB) would otherwise leave a later K = 100 call (80 KB) above the 64 KB default
    MP_LDS_ATTR(k_soft_blend, SOFT_MAX_K * 64 * 12 + 64 * (int)(sizeof(SoftTri) + sizeof(int)));
    hipLaunchKernelGGL(k_soft_blend, dim3(T), dim3(64), lds, st, verts, faces, colors, cam, H, W, sp, tx, offsets, list, image, sel);
    return (int)hipGetLastError();
}
