// HBM ceilings of the box as the MLP kernels see them: write-only, read-only and copy streams of 16-byte accesses
// (tools/bin/hbm_rate [GiB]).  Used by DESIGN.md: k_mlp_fwdsave writes 4 KiB per point, k_mlp_grad reads them back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_write(uint4* __restrict__ p, size_t n) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_read(const uint4* __restrict__ p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)), n = bytes / 16;
    uint4 *a, *b; unsigned* out;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 16, block = 256;
    auto time = [&](const char* name, double traffic, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f GB/s  (%.2f ms per %.1f GiB)\n", name, traffic * 5 / (ms * 1e-3) / 1e9, ms / 5, gib);
    };
    time("write 16 B/lane", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(block), 0, 0, a, n); });
    time("hipMemsetAsync", (double)bytes, [&] { hipMemsetAsync(a, 1, bytes, 0); });
    time("read 16 B/lane", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(block), 0, 0, a, n, out); });
    time("copy (read + write)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(block), 0, 0, a, b, n); });
    return 0;
}
