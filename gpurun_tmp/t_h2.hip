#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../multiply_amd/csrc/mlp_core.hpp"
using namespace mp;
__global__ void k(const float* a, float* o, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * t + 1 >= n) return;
    h2 z = to_h2(a[2 * t], a[2 * t + 1]);
    h2 e = exp2_neg_abs_h2(z);
    h2 l = log2_h2(e + (h2){(op_t)1.0f, (op_t)1.0f});
    h2 h = softplus2(z);
    h2 s = exp2_h2(z - h);
    h2 r = relu_h2(z);
    float* q = o + 12 * t;
    q[0] = (float)e[0]; q[1] = (float)e[1]; q[2] = (float)l[0]; q[3] = (float)l[1]; q[4] = (float)h[0]; q[5] = (float)h[1];
    q[6] = (float)s[0]; q[7] = (float)s[1]; q[8] = (float)r[0]; q[9] = (float)r[1]; q[10] = (float)z[0]; q[11] = (float)z[1];
}
int main() {
    const int n = 64;
    float ha[n], ho[6 * n];
    for (int i = 0; i < n; ++i) ha[i] = (i - 32) * 1.7f + 0.3f;
    ha[0] = -500.f; ha[1] = 500.f; ha[2] = 0.f; ha[3] = 60000.f;
    float *da, *dout;
    hipMalloc(&da, sizeof(ha)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, n);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int t = 0; t < 8; ++t) {
        const float* q = ho + 12 * t;
        for (int j = 0; j < 2; ++j) {
            const double z = q[10 + j];
            printf("z %9.3f  exp2(-|z|) %10.5g (%10.5g)  log2(1+u) %9.5f (%9.5f)  softplus %9.4f (%9.4f)  sigmoid %8.5f (%8.5f) relu %8.3f\n", z, q[j],
                   exp2(-fabs(z)), q[2 + j], log2(1 + exp2(-fabs(z))), q[4 + j], fmax(z, 0) + log2(1 + exp2(-fabs(z))), q[6 + j],
                   1.0 / (1.0 + exp2(-z)), q[8 + j]);
        }
    }
    return 0;
}
