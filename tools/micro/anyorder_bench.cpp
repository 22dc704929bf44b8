// micro: does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950, and what does a kernel boundary cost?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.000001f + 0.5f;
    p[blockIdx.x * 256 + threadIdx.x] = v;
}
int main() {
    float *a, *b;
    hipMalloc(&a, 1 << 24); hipMalloc(&b, 1 << 24);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 64;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, st);
            for (int k = 0; k < 20; ++k) {
                hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, nullptr, nullptr, 0, a, iters);
                if (mode == 0) hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, nullptr, nullptr, 0, b, iters);
                if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, b, iters);
            }
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %d (%s): %.1f us per pair\n", mode, mode == 0 ? "A then B ordered" : mode == 1 ? "A then B any-order" : "A only", ms * 1e3 / 20);
        }
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
