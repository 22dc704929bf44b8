// Shared device/host helpers for libsehip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_raw;  // storage type of a bf16 element in HBM / LDS
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SEHIP_WAVE 64

// error plumbing (api.cpp)
int sehip_set_error(int code, const char* fmt, ...);
// records which kernel instantiation the last product launch used (read back by sehip_last_kernel(), for bench/profiles)
void sehip_note_kernel(const char* fmt, ...);
// 1 while the deterministic schedule is on (sehip_set_deterministic, include/sehip.h)
int sehip_deterministic(void);
#define SEHIP_CHECK_LAUNCH(name)                                                      \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) return sehip_set_error(-2, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)
#define SEHIP_REQUIRE(cond, ...)                                      \
    do {                                                              \
        if (!(cond)) return sehip_set_error(-1, __VA_ARGS__);         \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_raw u) { return __uint_as_float(((unsigned)u) << 16); }
__device__ __forceinline__ bf16_raw f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(bf16_raw, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blocks of NW waves; every thread gets the result. `red` is NW floats of LDS.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
