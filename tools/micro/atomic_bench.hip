// Micro-benchmark: the flush of a split-K weight gradient.  256 workgroups x 512 threads each add a [128][640] fp32 tile (160 values
// per lane, conv_wgrad_v3's accumulator layout) into a [256][2560] array shared by 32 splits.  Variants: device-scope atomics,
// workgroup-scope atomics into one partial array per XCC (s_getreg XCC_ID), plain stores to a per-split scratch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void flush_kernel(float* dW, int K, int Npad, int gx, int ntn, int* xcc_seen) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = wv & 3, nh = wv >> 2;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int xi = rr % gx, split = (rr / gx) * 8 + xcd;
    const int nt = xi % ntn, cc = xi / ntn;
    const int n0 = nt * 128;
    const int Ctot = K / 10;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (tid == 0) xcc_seen[blockIdx.x] = (int)xcc;
    float* base = dW;
    if (MODE == 1 || MODE == 3) base = dW + (size_t)xcc * Npad * K;          // one partial array per XCC
    if (MODE == 2) base = dW + (size_t)split * Npad * K;                      // one per split, plain stores
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int n = n0 + 64 * nh + ni * 16 + 4 * (lane >> 4);
            const int k = it * Ctot + cc * 64 + 16 * w + (lane & 15);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float* p = &base[(size_t)(n + u) * K + k];
                const float v = 1.0f + ni + it;
                if (MODE == 0) atomicAdd(p, v);
                else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 3) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *p = v;
            }
        }
}

__global__ void reduce_kernel(const float* parts, int nparts, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i * 4 >= n) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < nparts; ++p) {
        const float4 v = *reinterpret_cast<const float4*>(parts + p * n + i * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i * 4) = s;
}

int main() {
    const int Npad = 256, K = 2560, ntn = 2, gx = ntn * (K / 10 / 64), splits = 32, grid = gx * splits;
    const size_t n = (size_t)Npad * K;
    float *buf, *out; int* seen;
    CHECK(hipMalloc(&buf, n * 4 * 32)); CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&seen, grid * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) fn();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.1f us\n", name, ms / 20 * 1e3);
    };
    printf("grid %d x 512, %zu floats per array, %.1f MB added per launch\n", grid, n, grid * 512.0 * 160 * 4 / 1e6);
    run("device-scope atomics, one array", [&] { flush_kernel<0><<<grid, 512>>>(buf, K, Npad, gx, ntn, seen); });
    run("agent-scope builtin, array per XCC", [&] { flush_kernel<3><<<grid, 512>>>(buf, K, Npad, gx, ntn, seen); });
    run("workgroup-scope atomics, array per XCC", [&] { flush_kernel<1><<<grid, 512>>>(buf, K, Npad, gx, ntn, seen); });
    run("plain stores, array per split", [&] { flush_kernel<2><<<grid, 512>>>(buf, K, Npad, gx, ntn, seen); });
    run("reduce 8 partial arrays", [&] { reduce_kernel<<<(n / 4 + 255) / 256, 256>>>(buf, 8, n, out); });
    run("reduce 32 partial arrays", [&] { reduce_kernel<<<(n / 4 + 255) / 256, 256>>>(buf, 32, n, out); });
    // correctness of the per-XCC variant: zero, one launch, reduce, compare with the expected count
    CHECK(hipMemset(buf, 0, n * 4 * 32));
    flush_kernel<1><<<grid, 512>>>(buf, K, Npad, gx, ntn, seen);
    reduce_kernel<<<(n / 4 + 255) / 256, 256>>>(buf, 16, n, out);
    CHECK(hipDeviceSynchronize());
    float* h = (float*)malloc(n * 4); int* hs = (int*)malloc(grid * 4);
    CHECK(hipMemcpy(h, out, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hs, seen, grid * 4, hipMemcpyDeviceToHost));
    // element (n, k): tap it = k / Ctot, ni = (n % 64) / 16: value (1 + ni + it) * splits
    size_t bad = 0;
    for (int r = 0; r < Npad; ++r)
        for (int k = 0; k < K; ++k) {
            const float want = (1.0f + (r % 64) / 16 + k / (K / 10)) * splits;
            if (h[(size_t)r * K + k] != want) ++bad;
        }
    printf("workgroup-scope per-XCC result: %zu wrong of %zu\n", bad, n);
    int hist[16] = {0}, rrobin = 0;
    for (int b = 0; b < grid; ++b) { hist[hs[b] & 15]++; if ((hs[b] & 7) == (b & 7)) ++rrobin; }
    printf("XCC ids seen:"); for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
    printf("   blocks with xcc == blockIdx & 7: %d of %d\n", rrobin, grid);
    return 0;
}
