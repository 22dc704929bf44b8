// Shared by csrc/rbn.hip (DCUnet's BatchNorm + LeakyReLU) and csrc/dcunet.hip (the fused tail: last BatchNorm + 1x1 conv + mask):
// 8-channel bf16 pieces, per-workgroup partial sums and their reduction by one wave, launch sizes.
#pragma once
#include "common.h"

#define RBN_SLOPE 0.01f
#define RBN_MAX_BLOCKS 512

struct RChunk8 { float v[8]; };
__device__ __forceinline__ RChunk8 r_unpack8(uint4 u) {
    RChunk8 c;
    c.v[0] = bf2f((bf16_raw)(u.x & 0xffff)); c.v[1] = bf2f((bf16_raw)(u.x >> 16));
    c.v[2] = bf2f((bf16_raw)(u.y & 0xffff)); c.v[3] = bf2f((bf16_raw)(u.y >> 16));
    c.v[4] = bf2f((bf16_raw)(u.z & 0xffff)); c.v[5] = bf2f((bf16_raw)(u.z >> 16));
    c.v[6] = bf2f((bf16_raw)(u.w & 0xffff)); c.v[7] = bf2f((bf16_raw)(u.w >> 16));
    return c;
}
__device__ __forceinline__ uint4 r_pack8(const float* v) {
    return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

// per-block partial sums: part[blockIdx.x][a*C + channel], a < NS
template <int NS>
__device__ __forceinline__ void rbn_block_partials(float (&s)[NS][8], int nq, int C, float* __restrict__ part,
                                                   float* lds /* [4][NS*8][nq] */) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* out = part + (size_t)blockIdx.x * NS * C;
#pragma unroll
    for (int a = 0; a < NS; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = s[a][j];
            for (int o = 32; o >= nq; o >>= 1) v += __shfl_xor(v, o, 64);   // lanes that share a piece column (nq divides 64)
            if (lane < nq) lds[(w * NS * 8 + a * 8 + j) * nq + lane] = v;
        }
    __syncthreads();
    for (int i = tid; i < NS * 8 * nq; i += 256) {
        const float t = lds[i] + lds[NS * 8 * nq + i] + lds[2 * NS * 8 * nq + i] + lds[3 * NS * 8 * nq + i];
        const int aj = i / nq, q = i - aj * nq;
        out[(size_t)(aj >> 3) * C + q * 8 + (aj & 7)] = t;
    }
}

// sums over the blocks' partials by one wave (all loads issued before the first add, see cbn.hip)
template <int NS>
__device__ __forceinline__ void rbn_wave_reduce(const float* __restrict__ part, int nblk, int C, int c, double (&out)[NS]) {
    float v[RBN_MAX_BLOCKS / 64][NS];
#pragma unroll
    for (int t = 0; t < RBN_MAX_BLOCKS / 64; ++t) {
        const int b = (threadIdx.x & 63) + 64 * t;
        const float* p = part + (size_t)(b < nblk ? b : 0) * NS * C + c;
#pragma unroll
        for (int k = 0; k < NS; ++k) v[t][k] = p[(size_t)k * C];
    }
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
#pragma unroll
    for (int t = 0; t < RBN_MAX_BLOCKS / 64; ++t) {
        const bool live = (int)(threadIdx.x & 63) + 64 * t < nblk;
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += live ? (double)v[t][k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) out[k] = wave_sum_d(acc[k]);
}


static inline int rbn_stat_blocks(long rows, int C) {
    const int rpb = 256 / (C >> 3);
    long g = (rows + (long)rpb * 8 - 1) / ((long)rpb * 8);
    if (g > RBN_MAX_BLOCKS) g = RBN_MAX_BLOCKS;
    if (g < 1) g = 1;
    return (int)g;
}
static inline int rbn_apply_blocks(long rows, int C) {
    const int rpb = 256 / (C >> 3);
    long g = (rows + (long)rpb * 8 - 1) / ((long)rpb * 8);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}
