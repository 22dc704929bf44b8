// DCUnet's ComplexBatchNorm2d = two INDEPENDENT real BatchNorm2d (bn_re on the real part, bn_im on the imaginary part:
// src/model/dcunet.py:374-386, no whitening) fused with the LeakyReLU(0.01) that follows it in Encoder / Decoder
// (src/model/dcunet.py:8-50), forward and backward, on channels-last bf16 activations [rows][C].
//   C = 2*Cs: real half | imaginary half, Cs channels STORED per half of which the first Cr are real channels of the
//   model (31 / 62 complex channels are stored as 32 / 64: every row is then a whole number of 16-byte pieces and the
//   MFMA K chunks stay aligned).  The padding channels are written as exact zeros by the forward pass and get zero
//   gradients.
//   forward : stats (1 read) -> per-channel finalize -> apply + LeakyReLU (1 read, 1 write)
//   backward: reduce (2 reads) -> per-channel finalize -> apply (2 reads, 1 write)
// all HBM-bound with 16-byte accesses; one thread owns one 8-channel piece of a row for all its rows.
//
//   o = w (y - mean) rstd + b;  z = o > 0 ? o : 0.01 o
//   g = dz * (o > 0 ? 1 : 0.01);  db = sum g;  dw = sum g xh  (xh = (y - mean) rstd);
//   dy = w rstd (g - mean(g) - xh mean(g xh))
#include <stdlib.h>
#include "common.h"
#include "rbn.h"

__global__ __launch_bounds__(256) void rbn_stats_kernel(const bf16_raw* __restrict__ y, long rows, int C, float* __restrict__ part) {
    __shared__ float lds[4 * 2 * 8 * 32];        // [4 waves][NS * 8][nq <= 32]
    const int nq = C >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    float s[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[0][j] = 0.f; s[1][j] = 0.f; }
    const long stride = (long)gridDim.x * rpb;
    for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 4 * stride) {
        uint4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {    // four independent 16-byte loads in flight per thread
            const long r = r0 + k * stride;
            u[k] = make_uint4(0u, 0u, 0u, 0u);
            if (r < rows) u[k] = *reinterpret_cast<const uint4*>(y + r * C + q * 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const RChunk8 a = r_unpack8(u[k]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s[0][j] += a.v[j]; s[1][j] += a.v[j] * a.v[j]; }
        }
    }
    rbn_block_partials<2>(s, nq, C, part, lds);
}

// coef record per stored channel: scale (w rstd), shift (b - mean scale), mean, rstd
// one wave per stored channel c: half = c / Cs (0 bn_re, 1 bn_im), index = c % Cs (>= Cr: padding -> all zero)
__global__ void rbn_finalize_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ w_re, const float* __restrict__ b_re,
                                    const float* __restrict__ w_im, const float* __restrict__ b_im, float* __restrict__ rm_re,
                                    float* __restrict__ rv_re, float* __restrict__ rm_im, float* __restrict__ rv_im,
                                    long* __restrict__ nbt_re, long* __restrict__ nbt_im, long rows, int Cs, int Cr, float eps,
                                    float momentum, int training, float4* __restrict__ coef, const float* __restrict__ shift) {
    const int c = blockIdx.x, C = 2 * Cs;
    // shift[c]: a per-channel constant the producer left OUT of the stored tensor (the convolution's bias: BatchNorm cancels it, and
    // a bias 50x the signal would cost the bf16 tensor 5-6 of its 8 mantissa bits).  y_reference = y_stored + shift: the
    // normalisation is unchanged, the running mean tracks mean + shift, the inference mean is running_mean - shift.
    const float sh = shift ? shift[c] : 0.f;
    const int half = c / Cs, i = c - half * Cs;
    if (i >= Cr) {
        if (threadIdx.x == 0) coef[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float w = (half ? w_im : w_re)[i], b = (half ? b_im : b_re)[i];
    float* rm = half ? rm_im : rm_re;
    float* rv = half ? rv_im : rv_re;
    const float rmo = rm[i], rvo = rv[i];
    float mean, var;
    if (training) {
        double a[2];
        rbn_wave_reduce<2>(part, nblk, C, c, a);
        if (threadIdx.x != 0) return;
        const double n = (double)rows;
        const double m = a[0] / n;
        double v = a[1] / n - m * m;
        if (v < 0.0) v = 0.0;
        mean = (float)m; var = (float)v;
        rm[i] = rmo + momentum * (mean + sh - rmo);
        rv[i] = rvo + momentum * ((float)(v * n / (n - 1.0)) - rvo);   // nn.BatchNorm2d: running_var takes the UNBIASED variance
        if (i == 0) { long* nb = half ? nbt_im : nbt_re; if (nb) nb[0] += 1; }
    } else {
        if (threadIdx.x != 0) return;
        mean = rmo - sh; var = rvo;
    }
    const float rstd = 1.f / sqrtf(var + eps);
    coef[c] = make_float4(w * rstd, b - mean * w * rstd, mean, rstd);
}

__global__ __launch_bounds__(256) void rbn_apply_kernel(const bf16_raw* __restrict__ y, const float4* __restrict__ coef, long rows,
                                                        int C, bf16_raw* __restrict__ z) {
    const int nq = C >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float4 k = coef[q * 8 + j]; sc[j] = k.x; sh[j] = k.y; }
    const long stride = (long)gridDim.x * rpb;
    for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 4 * stride) {
        uint4 u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long r = r0 + k * stride;
            u[k] = make_uint4(0u, 0u, 0u, 0u);
            if (r < rows) u[k] = *reinterpret_cast<const uint4*>(y + r * C + q * 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long r = r0 + k * stride;
            if (r >= rows) continue;
            const RChunk8 a = r_unpack8(u[k]);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = sc[j] * a.v[j] + sh[j];
                o[j] = v > 0.f ? v : RBN_SLOPE * v;
            }
            *reinterpret_cast<uint4*>(z + r * C + q * 8) = r_pack8(o);
        }
    }
}

// The finalize step inside the reduce launch (round 6, sehip_rbn_bwd_reduce_fin).  In the DCUnet step the weight gradients of the last
// decoder hold every CU's whole register file for 1-3 ms per workgroup; a NEW launch -- rbn_bwd_finalize_kernel's 2 Cs one-wave
// workgroups, microseconds of work -- waits for one of them to retire: 9 launches x ~105 us on the chain (profiles/r6_gaps_dcunet.txt).
// The reduce pass's workgroups are resident already: the LAST one to finish (ticket counter behind a device-scope release of its
// row of partials: cdna_hip_programming.md, slab-reducer recipe) adds the rows IN ROW ORDER -- deterministic whoever is last --
// and writes the parameter gradients and the apply pass's records.
struct RbnFin {
    unsigned* ticket;
    float *gw_re, *gb_re, *gw_im, *gb_im;
    float4* bcoef;
    int Cs, Cr;
};

// backward pass 1: per-channel sum g, sum g xh
template <bool FIN>
__global__ __launch_bounds__(256) void rbn_bwd_reduce_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ y,
                                                             const float4* __restrict__ coef, long rows, int C,
                                                             float* __restrict__ part, const RbnFin fin) {
    __shared__ float lds[4 * 2 * 8 * 32];        // [4 waves][NS * 8][nq <= 32]
    const int nq = C >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    float4 k[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = coef[q * 8 + j];
    float s[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[0][j] = 0.f; s[1][j] = 0.f; }
    const long stride = (long)gridDim.x * rpb;
    for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 2 * stride) {
        uint4 uy[2], ug[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long r = r0 + t * stride;
            uy[t] = make_uint4(0u, 0u, 0u, 0u); ug[t] = uy[t];
            if (r < rows) {
                uy[t] = *reinterpret_cast<const uint4*>(y + r * C + q * 8);
                ug[t] = *reinterpret_cast<const uint4*>(dz + r * C + q * 8);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const RChunk8 a = r_unpack8(uy[t]), g = r_unpack8(ug[t]);   // rows beyond the end: g == 0 contributes nothing
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float o = k[j].x * a.v[j] + k[j].y;
                const float gg = o > 0.f ? g.v[j] : RBN_SLOPE * g.v[j];
                s[0][j] += gg;
                s[1][j] += gg * (a.v[j] - k[j].z) * k[j].w;
            }
        }
    }
    rbn_block_partials<2>(s, nq, C, part, lds);
    if (!FIN) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the row has left
    __syncthreads();
    __shared__ unsigned is_last;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        is_last = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the layer's next call
    }
    __syncthreads();
    const int nblk = gridDim.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int half = c / fin.Cs, i = c - half * fin.Cs;
        if (i >= fin.Cr) { fin.bcoef[c] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        double a0 = 0.0, a1 = 0.0;
        for (int b0 = 0; b0 < nblk; b0 += 32) {            // 64 loads in flight per trip, rows added in row order
            float v0[32], v1[32];
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                const float* p = part + (size_t)(b0 + t < nblk ? b0 + t : 0) * 2 * C + c;
                v0[t] = p[0]; v1[t] = p[C];
            }
#pragma unroll
            for (int t = 0; t < 32; ++t)
                if (b0 + t < nblk) { a0 += (double)v0[t]; a1 += (double)v1[t]; }
        }
        const float4 k = coef[c];
        (half ? fin.gb_im : fin.gb_re)[i] = (float)a0;
        (half ? fin.gw_im : fin.gw_re)[i] = (float)a1;
        const double n = (double)rows;
        fin.bcoef[c] = make_float4(k.x, (float)(a0 / n), (float)(a1 / n), 0.f);
    }
}

// bcoef record per stored channel: a = w rstd, k1 = mean(g), k2 = mean(g xh)
__global__ void rbn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, const float4* __restrict__ coef, long rows, int Cs,
                                        int Cr, float* __restrict__ gw_re, float* __restrict__ gb_re, float* __restrict__ gw_im,
                                        float* __restrict__ gb_im, float4* __restrict__ bcoef) {
    const int c = blockIdx.x, C = 2 * Cs;
    const int half = c / Cs, i = c - half * Cs;
    if (i >= Cr) {
        if (threadIdx.x == 0) bcoef[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 k = coef[c];
    double a[2];
    rbn_wave_reduce<2>(part, nblk, C, c, a);
    if (threadIdx.x != 0) return;
    (half ? gb_im : gb_re)[i] = (float)a[0];
    (half ? gw_im : gw_re)[i] = (float)a[1];
    const double n = (double)rows;
    bcoef[c] = make_float4(k.x, (float)(a[0] / n), (float)(a[1] / n), 0.f);
}

__global__ __launch_bounds__(256) void rbn_bwd_apply_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ y,
                                                            const float4* __restrict__ coef, const float4* __restrict__ bcoef,
                                                            long rows, int C, bf16_raw* __restrict__ dy) {
    const int nq = C >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    float4 k[8], kb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { k[j] = coef[q * 8 + j]; kb[j] = bcoef[q * 8 + j]; }
    const long stride = (long)gridDim.x * rpb;
    for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 2 * stride) {
        uint4 uy[2], ug[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long r = r0 + t * stride;
            uy[t] = make_uint4(0u, 0u, 0u, 0u); ug[t] = uy[t];
            if (r < rows) {
                uy[t] = *reinterpret_cast<const uint4*>(y + r * C + q * 8);
                ug[t] = *reinterpret_cast<const uint4*>(dz + r * C + q * 8);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const long r = r0 + t * stride;
            if (r >= rows) continue;
            const RChunk8 a = r_unpack8(uy[t]), g = r_unpack8(ug[t]);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = k[j].x * a.v[j] + k[j].y;
                const float gg = v > 0.f ? g.v[j] : RBN_SLOPE * g.v[j];
                const float xh = (a.v[j] - k[j].z) * k[j].w;
                o[j] = kb[j].x * (gg - kb[j].y - xh * kb[j].z);
            }
            *reinterpret_cast<uint4*>(dy + r * C + q * 8) = r_pack8(o);
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int check_rbn(const char* who, long rows, int Cs, int Cr) {
    SEHIP_REQUIRE(rows > 1, "%s: BatchNorm needs more than one value per channel (rows=%ld)", who, rows);
    SEHIP_REQUIRE(Cs >= 8 && Cs <= 128 && (Cs & (Cs - 1)) == 0, "%s: stored channels per half Cs=%d must be 8, 16, 32, 64 or 128", who, Cs);
    SEHIP_REQUIRE(Cr >= 1 && Cr <= Cs, "%s: Cr=%d must be in [1, Cs=%d]", who, Cr, Cs);
    return 0;
}
extern "C" long sehip_rbn_scratch_floats(long rows, int Cs) { return (long)rbn_stat_blocks(rows, 2 * Cs) * 2L * 2 * Cs; }

extern "C" int sehip_rbn_stats(const void* y, long rows, int Cs, int Cr, float* part, void* stream) {
    if (int e = check_rbn("rbn_stats", rows, Cs, Cr)) return e;
    rbn_stats_kernel<<<rbn_stat_blocks(rows, 2 * Cs), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, rows, 2 * Cs, part);
    SEHIP_CHECK_LAUNCH("rbn_stats");
    return 0;
}

extern "C" int sehip_rbn_finalize_s(const float* part, const float* w_re, const float* b_re, const float* w_im, const float* b_im,
                                    float* rm_re, float* rv_re, float* rm_im, float* rv_im, long* nbt_re, long* nbt_im, long rows,
                                    int Cs, int Cr, float eps, float momentum, int training, const float* shift, float* coef,
                                    void* stream) {
    if (int e = check_rbn("rbn_finalize", rows, Cs, Cr)) return e;
    rbn_finalize_kernel<<<2 * Cs, 64, 0, (hipStream_t)stream>>>(part, rbn_stat_blocks(rows, 2 * Cs), w_re, b_re, w_im, b_im, rm_re, rv_re,
                                                             rm_im, rv_im, nbt_re, nbt_im, rows, Cs, Cr, eps, momentum, training,
                                                             (float4*)coef, shift);
    SEHIP_CHECK_LAUNCH("rbn_finalize");
    return 0;
}
extern "C" int sehip_rbn_finalize(const float* part, const float* w_re, const float* b_re, const float* w_im, const float* b_im,
                                  float* rm_re, float* rv_re, float* rm_im, float* rv_im, long* nbt_re, long* nbt_im, long rows,
                                  int Cs, int Cr, float eps, float momentum, int training, float* coef, void* stream) {
    return sehip_rbn_finalize_s(part, w_re, b_re, w_im, b_im, rm_re, rv_re, rm_im, rv_im, nbt_re, nbt_im, rows, Cs, Cr, eps, momentum,
                                training, nullptr, coef, stream);
}

extern "C" int sehip_rbn_apply(const void* y, const float* coef, long rows, int Cs, int Cr, void* z, void* stream) {
    if (int e = check_rbn("rbn_apply", rows, Cs, Cr)) return e;
    rbn_apply_kernel<<<rbn_apply_blocks(rows, 2 * Cs), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, (const float4*)coef, rows,
                                                                                   2 * Cs, (bf16_raw*)z);
    SEHIP_CHECK_LAUNCH("rbn_apply");
    return 0;
}

extern "C" int sehip_rbn_bwd_reduce(const void* dz, const void* y, const float* coef, long rows, int Cs, int Cr, float* part,
                                    void* stream) {
    if (int e = check_rbn("rbn_bwd_reduce", rows, Cs, Cr)) return e;
    rbn_bwd_reduce_kernel<false><<<rbn_stat_blocks(rows, 2 * Cs), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)dz, (const bf16_raw*)y,
                                                                                              (const float4*)coef, rows, 2 * Cs, part, RbnFin{});
    SEHIP_CHECK_LAUNCH("rbn_bwd_reduce");
    return 0;
}

extern "C" int sehip_rbn_bwd_reduce_fin(const void* dz, const void* y, const float* coef, long rows, int Cs, int Cr, float* part,
                                        unsigned* ticket, float* gw_re, float* gb_re, float* gw_im, float* gb_im, float* bcoef,
                                        void* stream) {
    if (int e = check_rbn("rbn_bwd_reduce_fin", rows, Cs, Cr)) return e;
    SEHIP_REQUIRE(ticket && bcoef && part, "rbn_bwd_reduce_fin: null ticket / bcoef / part");
    const RbnFin fin{ticket, gw_re, gb_re, gw_im, gb_im, (float4*)bcoef, Cs, Cr};
    rbn_bwd_reduce_kernel<true><<<rbn_stat_blocks(rows, 2 * Cs), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)dz, (const bf16_raw*)y,
                                                                                             (const float4*)coef, rows, 2 * Cs, part, fin);
    SEHIP_CHECK_LAUNCH("rbn_bwd_reduce_fin");
    return 0;
}

extern "C" int sehip_rbn_bwd_finalize(const float* part, const float* coef, long rows, int Cs, int Cr, float* gw_re, float* gb_re,
                                      float* gw_im, float* gb_im, float* bcoef, void* stream) {
    if (int e = check_rbn("rbn_bwd_finalize", rows, Cs, Cr)) return e;
    rbn_bwd_finalize_kernel<<<2 * Cs, 64, 0, (hipStream_t)stream>>>(part, rbn_stat_blocks(rows, 2 * Cs), (const float4*)coef, rows, Cs, Cr,
                                                                 gw_re, gb_re, gw_im, gb_im, (float4*)bcoef);
    SEHIP_CHECK_LAUNCH("rbn_bwd_finalize");
    return 0;
}

extern "C" int sehip_rbn_bwd_apply(const void* dz, const void* y, const float* coef, const float* bcoef, long rows, int Cs, int Cr,
                                   void* dy, void* stream) {
    if (int e = check_rbn("rbn_bwd_apply", rows, Cs, Cr)) return e;
    rbn_bwd_apply_kernel<<<rbn_apply_blocks(rows, 2 * Cs), 256, 0, (hipStream_t)stream>>>(
        (const bf16_raw*)dz, (const bf16_raw*)y, (const float4*)coef, (const float4*)bcoef, rows, 2 * Cs, (bf16_raw*)dy);
    SEHIP_CHECK_LAUNCH("rbn_bwd_apply");
    return 0;
}
