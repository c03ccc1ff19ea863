// Input producer: sub-pixel sampling of a resident frame.
// Replaces weighted_sampling + bilinear_interpolation of the reference's dataset (code/lib/datasets/Hi4D.py:8-20,
// 59-88), which decode the frame's PNGs and gather on the CPU for every item.  Here the sequence's frames stay in HBM
// as bytes (1 MP x 3 B per frame: a 300-frame sequence is < 1 GB of 288) and one launch gathers an item's samples.
// HBM-bound byte work, 4 neighbouring pixels per sample and channel; the arithmetic is done in double like the
// reference's numpy code and rounded once to fp32 (Hi4D.py:274, 286: .astype(np.float32)).
#include <hip/hip_runtime.h>
#include "../../include/multiply_hip.h"

namespace {

// One thread per sample.  pos = (row, col) in pixels, row in [0, H-1), col in [0, W-1)  (Hi4D.py:66-72).
// bilinear_interpolation(xs, ys, map) = [x2-xs, xs-x1] . [[m(x1,y1), m(x1,y2)], [m(x2,y1), m(x2,y2)]] . [y2-ys, ys-y1]^T
__global__ void k_sample_pixels(const unsigned char* __restrict__ img, const unsigned char* __restrict__ mask,
                                const float* __restrict__ extra, int n_extra, const double* __restrict__ pos, int n, int H,
                                int W, float* __restrict__ rgb, float* __restrict__ uv, float* __restrict__ mask_out,
                                float* __restrict__ extra_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double xs = pos[2 * i], ys = pos[2 * i + 1];
    int x1 = (int)floor(xs), y1 = (int)floor(ys);
    x1 = min(max(x1, 0), H - 2);   // the reference indexes x1 + 1 unchecked; positions are < H-1 by construction
    y1 = min(max(y1, 0), W - 2);
    const int x2 = x1 + 1, y2 = y1 + 1;
    const double wx1 = (double)x2 - xs, wx2 = xs - (double)x1, wy1 = (double)y2 - ys, wy2 = ys - (double)y1;
    const size_t p11 = (size_t)x1 * W + y1, p12 = (size_t)x1 * W + y2, p21 = (size_t)x2 * W + y1, p22 = (size_t)x2 * W + y2;
    // (dx @ Q @ dy): first contract the rows with dx, then the columns with dy -- the order numpy's matmul chain uses
    auto lerp = [&](double q11, double q12, double q21, double q22) {
        const double a = wx1 * q11 + wx2 * q21, b = wx1 * q12 + wx2 * q22;
        return a * wy1 + b * wy2;
    };
    if (rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c)   // img / 255 in double (Hi4D.py:232), then interpolated
            rgb[3 * i + c] = (float)lerp(img[3 * p11 + c] / 255.0, img[3 * p12 + c] / 255.0, img[3 * p21 + c] / 255.0,
                                         img[3 * p22 + c] / 255.0);
    }
    if (uv) {   // uv[r][c] = (c, r)  (Hi4D.py:254-255: mgrid flipped to x, y order)
        uv[2 * i] = (float)lerp((double)y1, (double)y2, (double)y1, (double)y2);
        uv[2 * i + 1] = (float)lerp((double)x1, (double)x1, (double)x2, (double)x2);
    }
    if (mask_out) mask_out[i] = (float)lerp((double)mask[p11], (double)mask[p12], (double)mask[p21], (double)mask[p22]);
    for (int c = 0; c < n_extra; ++c)   // e.g. the SAM mask (H, W, P) fp32 (Hi4D.py:266-267)
        extra_out[(size_t)i * n_extra + c] =
            (float)lerp((double)extra[p11 * n_extra + c], (double)extra[p12 * n_extra + c], (double)extra[p21 * n_extra + c],
                        (double)extra[p22 * n_extra + c]);
}

}  // namespace

extern "C" int mp_sample_pixels(const unsigned char* img, const unsigned char* mask, const float* extra, int n_extra,
                                const double* pos, int n, int H, int W, float* rgb, float* uv, float* mask_out,
                                float* extra_out, void* stream) {
    if (n <= 0) return 0;
    if (H < 2 || W < 2 || n_extra < 0 || (n_extra > 0 && (!extra || !extra_out)) || (rgb && !img) || (mask_out && !mask))
        return -1;
    hipLaunchKernelGGL(k_sample_pixels, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, img, mask, extra, n_extra,
                       pos, n, H, W, rgb, uv, mask_out, extra_out);
    return (int)hipGetLastError();
}
