// Demucs (src/model/demucs.py:272-501), everything that is not a convolution / linear product (those are sehip_gemm products of the
// implicit-GEMM engine over channels-last bf16 activations [B][T][C]):
//   prep / post   normalise + pad + julius-style x2 up-sampling, and /2 down-sampling + de-normalise + center_trim (:453-470, :485-490)
//   gn / act      GroupNorm (:176, :382) fused with GELU or GLU, LayerScale (:52-71) and the residual / skip additions (:204-207, :483)
//   lstm          a bidirectional nn.LSTM layer (:83): ONE persistent launch whose workgroups hand h(t) / the gate gradients over once
//                 per time step (dmx_lstm_seq_*), or one launch per time step (dmx_lstm_step_*: fallback and cross-check)
//   frames        the BLSTM's overlapping chunks of max_steps frames (:91-117): gather, middle-part pick and their adjoints
//   attn          LocalState (:210-269, nfreqs = 0): scores, distance penalty, softmax over the key axis, weighted content
// Thread layout of the streaming kernels: a thread owns ONE piece of 8 output channels for all its frames (the launch's thread
// count is a multiple of the number of pieces, so piece = global thread id % pieces), hence per-channel affine terms and the
// per-channel gradient sums stay in registers.
#ifndef DMX_FU
#define DMX_FU 2      // frames per trip of the activation passes (loads of all of them in flight)
#endif
#include "common.h"
#include "det.h"
float* sehip_wgrad_scratch(hipStream_t st, size_t bytes);   // csrc/wgrad3.hip: per-stream pool of partial arrays
#include <math.h>
#include <stdlib.h>

namespace {

struct D8 { float v[8]; };
__device__ __forceinline__ D8 ld8(const bf16_raw* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    D8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void st8(bf16_raw* p, const float (&v)[8]) {
    uint4 u;
    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void ld8f(const float* __restrict__ p, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[j];
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
// exact GELU (nn.GELU default, approximate='none') and its derivative
__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// piece walk: piece index i = t * nq + q; this thread's q is fixed, t = t0, t0 + tstep, ...
struct Walk { int q, t0, tstep; };
__device__ __forceinline__ Walk walk(int nq) {
    const long gi = (long)blockIdx.x * 256 + threadIdx.x;
    Walk w;
    w.q = (int)(gi % nq);
    w.t0 = (int)(gi / nq);
    w.tstep = (int)(((long)gridDim.x * 256) / nq);
    return w;
}

__device__ __forceinline__ void moments(const double* __restrict__ stats, int b, int G, int g, double n, float eps, float& mu, float& rs) {
    const double s = stats[(b * G + g) * 2], q = stats[(b * G + g) * 2 + 1];
    const double m = s / n;
    double var = q / n - m * m;
    if (var < 0) var = 0;
    mu = (float)m;
    rs = (float)(1.0 / sqrt(var + (double)eps));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// prep / post
// ---------------------------------------------------------------------------------------------------------------------------
// acc[b] += (sum, sum of squares) of the mono mix over this workgroup's share of the clip (double atomics; the caller zeroes acc)
__global__ __launch_bounds__(256) void dmx_moments_kernel(const float* __restrict__ mix, int ac, int T, double* __restrict__ acc, const DetCtx dc) {
    const int b = blockIdx.y;
    __shared__ double red[2][4];
    double s = 0, q = 0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < T; t += (long)gridDim.x * 256) {
        float m = 0.f;
        for (int a = 0; a < ac; ++a) m += mix[((long)b * ac + a) * T + t];
        m /= (float)ac;
        s += m; q += (double)m * m;
    }
    s = wave_sum_d(s); q = wave_sum_d(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    __shared__ double pair[2];
    if (threadIdx.x == 0) {
        pair[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        pair[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    __syncthreads();
    det_group_add(pair, 2, acc + 2 * b, dc, b, blockIdx.x, gridDim.x);      // (deterministic schedule: slots added in slot order, csrc/det.h)
}
// ms[b] = (mean, unbiased std) of the mono mix (src/model/demucs.py:457-461); normalize == 0: (0, 1)
__global__ void dmx_moments_finish_kernel(const double* __restrict__ acc, int B, int T, int normalize, float* __restrict__ ms) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (!normalize) { ms[2 * b] = 0.f; ms[2 * b + 1] = 1.f; return; }
    const double s = acc[2 * b], q = acc[2 * b + 1];
    const double mean = s / T;
    double var = (q - s * mean) / (T > 1 ? T - 1 : 1);
    if (var < 0) var = 0;
    ms[2 * b] = (float)mean;
    ms[2 * b + 1] = (float)sqrt(var);
}

// x[b][u][c] (bf16, acp channels, channels >= ac are zero).  up == 0: u indexes the padded signal; up == 1: u = 2 n + i and
// x = sum_j kup[i][j] xin(clamp(n + j - width)), xin = the normalised, zero-padded signal of Tv samples (replicate at the ends)
__global__ __launch_bounds__(256) void dmx_prep_kernel(const float* __restrict__ mix, const float* __restrict__ ms, const float* __restrict__ kup,
                                                       int ac, int acp, int T, int padl, int Tv, int up, int width, int KL,
                                                       bf16_raw* __restrict__ x) {
    const int b = blockIdx.y;
    const long T0 = up ? 2L * Tv : Tv;
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= T0) return;
    const float mean = ms[2 * b], inv = 1.f / (1e-5f + ms[2 * b + 1]);
    for (int c = 0; c < acp; ++c) {
        float v = 0.f;
        if (c < ac) {
            const float* src = mix + ((long)b * ac + c) * T;
            if (!up) {
                const long p = u - padl;
                v = (p >= 0 && p < T) ? (src[p] - mean) * inv : 0.f;
            } else {
                const long n = u >> 1;
                const float* k = kup + (u & 1) * KL;
                for (int j = 0; j < KL; ++j) {
                    long p = n + j - width;
                    p = p < 0 ? 0 : (p >= Tv ? Tv - 1 : p);
                    p -= padl;
                    if (p >= 0 && p < T) v += k[j] * (src[p] - mean) * inv;
                }
            }
        }
        x[((long)b * T0 + u) * acp + c] = f2bf(v);
    }
}

// out[b][c][n] = z(n + padl) * std + mean, z = y (down == 0) or y down-sampled by 2: z[m] = sum_j kdn[j] y(clamp(2 m + j - width))
__global__ __launch_bounds__(256) void dmx_post_kernel(const float* __restrict__ y /*[B][Tf][cop]*/, const float* __restrict__ ms,
                                                       const float* __restrict__ kdn, int co, int cop, long Tf, int padl, int T, int down,
                                                       int width, int KL, float* __restrict__ out /*[B][co][T]*/) {
    const int b = blockIdx.y;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= T) return;
    const float mean = ms[2 * b], sd = ms[2 * b + 1];
    const float* yb = y + (long)b * Tf * cop;
    const long m = n + padl;
    for (int c = 0; c < co; ++c) {
        float v = 0.f;
        if (!down) {
            v = yb[m * cop + c];
        } else {
            for (int j = 0; j < KL; ++j) {
                long p = 2 * m + j - width;
                p = p < 0 ? 0 : (p >= Tf ? Tf - 1 : p);
                v += kdn[j] * yb[p * cop + c];
            }
        }
        out[((long)b * co + c) * T + n] = v * sd + mean;
    }
}

// dy[b][u][c] (bf16, cop channels) = std * sum over the output samples the position fed
__global__ __launch_bounds__(256) void dmx_post_bwd_kernel(const float* __restrict__ dout /*[B][co][T]*/, const float* __restrict__ ms,
                                                           const float* __restrict__ kdn, int co, int cop, long Tf, int padl, int T, int down,
                                                           int width, int KL, bf16_raw* __restrict__ dy) {
    const int b = blockIdx.y;
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= Tf) return;
    const float sd = ms[2 * b + 1];
    for (int c = 0; c < cop; ++c) {
        float v = 0.f;
        if (c < co) {
            const float* d = dout + ((long)b * co + c) * T;
            if (!down) {
                const long n = u - padl;
                if (n >= 0 && n < T) v = d[n];
            } else {
                // padded positions p with clamp(p) == u
                long plo = u, phi = u;
                if (u == 0) plo = -width;
                if (u == Tf - 1) phi = Tf + width + 1;
                for (long p = plo; p <= phi; ++p) {
                    // j = p + width - 2 m in [0, KL)  <=>  m in [(p + width - KL + 1) / 2 (ceil), (p + width) / 2 (floor)]
                    long mlo = p + width - KL + 1;
                    mlo = mlo <= 0 ? 0 : (mlo + 1) / 2;
                    long mhi = (p + width) / 2;
                    if (mlo < padl) mlo = padl;
                    if (mhi > padl + T - 1) mhi = padl + T - 1;
                    for (long m = mlo; m <= mhi; ++m) v += kdn[p + width - 2 * m] * d[m - padl];
                }
            }
        }
        dy[((long)b * Tf + u) * cop + c] = f2bf(v * sd);
    }
}

// The resampling paths of the three kernels above through LDS: a workgroup stages the window of its 256 positions once per channel
// (channel-major, so that the stride-2 reads of the decimator are 2-way bank conflicts at most) and the taps run from LDS with
// the filter coefficient as a scalar load.  The per-thread versions gather every tap from global memory at a stride of
// 2 cop floats: 104 taps x 3 M positions of uncoalesced 4-byte loads, 287 / 214 / 128 us at the C3 shape.
__global__ __launch_bounds__(256) void dmx_prep_up_kernel(const float* __restrict__ mix, const float* __restrict__ ms, const float* __restrict__ kup,
                                                          int ac, int acp, int T, int padl, int Tv, int width, int KL,
                                                          bf16_raw* __restrict__ x) {
    extern __shared__ float win[];                       // [ac][128 + KL]
    const int b = blockIdx.y;
    const long T0 = 2L * Tv;
    const long u0 = (long)blockIdx.x * 256;              // even
    const long n0 = u0 >> 1;
    const int WL = 128 + KL;
    const float mean = ms[2 * b], inv = 1.f / (1e-5f + ms[2 * b + 1]);
    for (int i = threadIdx.x; i < ac * WL; i += 256) {
        const int c = i / WL, idx = i - c * WL;
        long p = n0 + idx - width;
        p = p < 0 ? 0 : (p >= Tv ? Tv - 1 : p);
        p -= padl;
        win[i] = (p >= 0 && p < T) ? (mix[((long)b * ac + c) * T + p] - mean) * inv : 0.f;
    }
    __syncthreads();
    const long u = u0 + threadIdx.x;
    if (u >= T0) return;
    const int nl = threadIdx.x >> 1;
    const float* k = kup + (threadIdx.x & 1) * KL;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.f;
    for (int c = 0; c < ac && c < 8; ++c) {
        const float* wc = win + c * WL + nl;
        float acc = 0.f;
        for (int j = 0; j < KL; ++j) acc += k[j] * wc[j];
        v[c] = acc;
    }
    bf16_raw* dst = x + ((long)b * T0 + u) * acp;
    for (int c = 0; c < acp; ++c) dst[c] = f2bf(c < 8 ? v[c] : 0.f);
}

__global__ __launch_bounds__(256) void dmx_post_down_kernel(const float* __restrict__ y /*[B][Tf][cop]*/, const float* __restrict__ ms,
                                                            const float* __restrict__ kdn, int co, int cop, long Tf, int padl, int T,
                                                            int width, int KL, float* __restrict__ out /*[B][co][T]*/) {
    extern __shared__ float win[];                       // [co][512 + KL]
    const int b = blockIdx.y;
    const long nb = (long)blockIdx.x * 256;
    const int WL = 512 + KL;
    const float* yb = y + (long)b * Tf * cop;
    const long p0 = 2 * (nb + padl) - width;
    for (int i = threadIdx.x; i < WL; i += 256) {
        long p = p0 + i;
        p = p < 0 ? 0 : (p >= Tf ? Tf - 1 : p);
        for (int c = 0; c < co; ++c) win[c * WL + i] = yb[p * cop + c];
    }
    __syncthreads();
    const long n = nb + threadIdx.x;
    if (n >= T) return;
    const float mean = ms[2 * b], sd = ms[2 * b + 1];
    for (int c = 0; c < co; ++c) {
        const float* wc = win + c * WL + 2 * threadIdx.x;
        float acc = 0.f;
        for (int j = 0; j < KL; ++j) acc += kdn[j] * wc[j];
        out[((long)b * co + c) * T + n] = acc * sd + mean;
    }
}

__global__ __launch_bounds__(256) void dmx_post_bwd_down_kernel(const float* __restrict__ dout /*[B][co][T]*/, const float* __restrict__ ms,
                                                                const float* __restrict__ kdn, int co, int cop, long Tf, int padl, int T,
                                                                int width, int KL, bf16_raw* __restrict__ dy) {
    extern __shared__ float win[];                       // [co][WL]: d[m - padl] for m in [mbase, mbase + WL)
    const int b = blockIdx.y;
    const long u0 = (long)blockIdx.x * 256;
    const int WL = 128 + KL / 2 + 4;
    long mbase = u0 + width - KL + 1;                    // smallest m any interior position of the block needs: ceil(. / 2)
    mbase = mbase <= 0 ? 0 : (mbase + 1) / 2;
    for (int i = threadIdx.x; i < co * WL; i += 256) {
        const int c = i / WL, idx = i - c * WL;
        const long m = mbase + idx;
        win[i] = (m >= padl && m <= padl + T - 1) ? dout[((long)b * co + c) * T + (m - padl)] : 0.f;
    }
    __syncthreads();
    const long u = u0 + threadIdx.x;
    if (u >= Tf) return;
    const float sd = ms[2 * b + 1];
    bf16_raw* dst = dy + ((long)b * Tf + u) * cop;
    for (int c = 0; c < cop; ++c) {
        float v = 0.f;
        if (c < co) {
            if (u == 0 || u == Tf - 1) {                 // the clamped ends collect the padded positions too: the general loop
                const float* d = dout + ((long)b * co + c) * T;
                long plo = u, phi = u;
                if (u == 0) plo = -width;
                if (u == Tf - 1) phi = Tf + width + 1;
                for (long p = plo; p <= phi; ++p) {
                    long mlo = p + width - KL + 1;
                    mlo = mlo <= 0 ? 0 : (mlo + 1) / 2;
                    long mhi = (p + width) / 2;
                    if (mlo < padl) mlo = padl;
                    if (mhi > padl + T - 1) mhi = padl + T - 1;
                    for (long m = mlo; m <= mhi; ++m) v += kdn[p + width - 2 * m] * d[m - padl];
                }
            } else {
                long mlo = u + width - KL + 1;
                mlo = mlo <= 0 ? 0 : (mlo + 1) / 2;
                const long mhi = (u + width) / 2;
                const float* wc = win + c * WL;
                for (long m = mlo; m <= mhi; ++m) v += kdn[u + width - 2 * m] * wc[m - mbase];     // (rows outside the clip are zeros)
            }
        }
        dst[c] = f2bf(v * sd);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// GroupNorm + activation family.  mode 0: z = gelu(n(y)) over C channels; mode 1: z = n(y)[:C/2] * sigmoid(n(y)[C/2:]) (GLU).
// Optional LayerScale + residual: z <- resid + scale[c] * z; optional addend: z <- z + add.  n = GroupNorm(G) or identity.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dmx_gn_stats_kernel(const bf16_raw* __restrict__ y, int T, int C, int G, double* __restrict__ stats,
                                                           const DetCtx dc) {
    const int b = blockIdx.y, nq = C >> 3;
    const Walk w = walk(nq);
    const int g = (w.q * 8) / (C / G);
    __shared__ float acc[16];
    if (threadIdx.x < 16) acc[threadIdx.x] = 0.f;
    __syncthreads();
    const bf16_raw* base = y + (long)b * T * C + w.q * 8;
    float s = 0.f, q = 0.f;
    for (int t = w.t0; t < T; t += w.tstep) {
        const D8 x = ld8(base + (long)t * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += x.v[j]; q += x.v[j] * x.v[j]; }
    }
    if (dc.part == nullptr) {
        atomicAdd(&acc[2 * g], s);
        atomicAdd(&acc[2 * g + 1], q);
        __syncthreads();
    } else {         // deterministic schedule: one thread per group adds the threads' sums in thread order, the workgroups in slot order
        __shared__ float tmpv[256];
        __shared__ int tmps[256];
        det_slot_sum(s, 2 * g, 2 * G, acc, tmpv, tmps, false);
        det_slot_sum(q, 2 * g + 1, 2 * G, acc, tmpv, tmps, true);
    }
    det_group_add(acc, 2 * G, stats + (long)b * G * 2, dc, b, blockIdx.x, gridDim.x);
}

template <int MODE>
__global__ __launch_bounds__(256) void dmx_act_fwd_kernel(const bf16_raw* __restrict__ y, const double* __restrict__ stats,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, int G, float eps,
                                                          const float* __restrict__ scale, const bf16_raw* __restrict__ resid,
                                                          const bf16_raw* __restrict__ add, int T, int C, bf16_raw* __restrict__ out) {
    const int b = blockIdx.y;
    const int Co = MODE ? C >> 1 : C, nq = Co >> 3;
    const Walk w = walk(nq);
    const int c0 = w.q * 8, c1 = (C >> 1) + c0;
    const bool norm = stats != nullptr;
    float ga[8], ba[8], gg[8], bg[8], sc[8];
    float mua = 0.f, rsa = 1.f, mug = 0.f, rsg = 1.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = gg[j] = 1.f; ba[j] = bg[j] = 0.f; sc[j] = 1.f; }
    if (norm) {
        const double n = (double)T * (C / G);
        ld8f(gamma + c0, ga); ld8f(beta + c0, ba);
        moments(stats, b, G, c0 / (C / G), n, eps, mua, rsa);
        if (MODE) { ld8f(gamma + c1, gg); ld8f(beta + c1, bg); moments(stats, b, G, c1 / (C / G), n, eps, mug, rsg); }
    }
    if (scale) ld8f(scale + c0, sc);
    const bf16_raw* yb = y + (long)b * T * C;
    // DMX_FU frames per trip, every load of them requested before the first use (one frame per trip: 3.0 TB/s on the 300-MB tensors)
    for (int t0 = w.t0; t0 < T; t0 += DMX_FU * w.tstep) {
        D8 a[DMX_FU], g[DMX_FU], r1[DMX_FU], r2[DMX_FU];
        bool ok[DMX_FU];
#pragma unroll
        for (int u = 0; u < DMX_FU; ++u) {
            const int t = t0 + u * w.tstep;
            ok[u] = t < T;
            const int tt = ok[u] ? t : t0;
            a[u] = ld8(yb + (long)tt * C + c0);
            if (MODE) g[u] = ld8(yb + (long)tt * C + c1);
            const long o = ((long)b * T + tt) * Co + c0;
            if (scale) r1[u] = ld8(resid + o);
            if (add) r2[u] = ld8(add + o);
        }
#pragma unroll
        for (int u = 0; u < DMX_FU; ++u) {
            if (!ok[u]) continue;
            const int t = t0 + u * w.tstep;
            float v[8];
            if (MODE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = ((a[u].v[j] - mua) * rsa * ga[j] + ba[j]) * sigm((g[u].v[j] - mug) * rsg * gg[j] + bg[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu((a[u].v[j] - mua) * rsa * ga[j] + ba[j]);
            }
            if (scale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = r1[u].v[j] + sc[j] * v[j];
            }
            if (add) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += r2[u].v[j];
            }
            st8(out + ((long)b * T + t) * Co + c0, v);
        }
    }
}

// pass 1 of the backward (GroupNorm present): sums[b][g] += (sum dyh, sum dyh xh), gch += {dgamma [C] | dbeta [C] | dscale [Co]}
template <int MODE>
__global__ __launch_bounds__(256) void dmx_act_bwd_reduce_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ y,
                                                                 const double* __restrict__ stats, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int G, float eps,
                                                                 const float* __restrict__ scale, int T, int C, double* __restrict__ sums,
                                                                 float* __restrict__ gch, float* __restrict__ part, const DetCtx dc) {
    // part != NULL: the block's 2 C + Co per-channel sums go to ITS ROW of `part` by plain stores and the apply pass adds the rows
    // into gch (dmx_colsum_share) -- flushing every block with fp32 atomics was up to 2 048 blocks x 320-10 240 addresses per launch:
    // a ~26-us floor under each of the 32 launches of a step, whatever the tensor's size
    extern __shared__ float lds[];   // [2 C + Co] per-channel partials of the block, then [16] group partials
    const int b = blockIdx.y;
    const int Co = MODE ? C >> 1 : C, nq = Co >> 3;
    const int NV = 2 * C + Co;
    for (int i = threadIdx.x; i < NV + 16; i += 256) lds[i] = 0.f;
    __syncthreads();
    const Walk w = walk(nq);
    const int c0 = w.q * 8, c1 = (C >> 1) + c0;
    const int grp_a = c0 / (C / G), grp_g = c1 / (C / G);
    float ga[8], ba[8], gg[8], bg[8], sc[8];
    float mua, rsa, mug = 0.f, rsg = 1.f;
    const double n = (double)T * (C / G);
    ld8f(gamma + c0, ga); ld8f(beta + c0, ba);
    moments(stats, b, G, grp_a, n, eps, mua, rsa);
#pragma unroll
    for (int j = 0; j < 8; ++j) { gg[j] = 1.f; bg[j] = 0.f; sc[j] = 1.f; }
    if (MODE) { ld8f(gamma + c1, gg); ld8f(beta + c1, bg); moments(stats, b, G, grp_g, n, eps, mug, rsg); }
    if (scale) ld8f(scale + c0, sc);
    float dga[8], dba[8], dgg[8], dbg[8], dsc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dga[j] = dba[j] = dgg[j] = dbg[j] = dsc[j] = 0.f;
    float s1a = 0.f, s2a = 0.f, s1g = 0.f, s2g = 0.f;
    const bf16_raw* yb = y + (long)b * T * C;
    for (int t0 = w.t0; t0 < T; t0 += DMX_FU * w.tstep) {       // DMX_FU frames per trip, their loads in flight together
      D8 a_[DMX_FU], d_[DMX_FU], g_[DMX_FU];
      bool ok_[DMX_FU];
#pragma unroll
      for (int u = 0; u < DMX_FU; ++u) {
          const int t = t0 + u * w.tstep;
          ok_[u] = t < T;
          const int tt = ok_[u] ? t : t0;
          a_[u] = ld8(yb + (long)tt * C + c0);
          d_[u] = ld8(dz + ((long)b * T + tt) * Co + c0);
          if (MODE) g_[u] = ld8(yb + (long)tt * C + c1);
      }
#pragma unroll
      for (int u = 0; u < DMX_FU; ++u) {
        if (!ok_[u]) continue;
        const D8 a = a_[u], d = d_[u];
        if (MODE) {
            const D8 g = g_[u];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xa = (a.v[j] - mua) * rsa, xg = (g.v[j] - mug) * rsg;
                const float na = xa * ga[j] + ba[j], sg = sigm(xg * gg[j] + bg[j]);
                dsc[j] += d.v[j] * na * sg;
                const float dv = d.v[j] * sc[j];
                const float dna = dv * sg, dng = dv * na * sg * (1.f - sg);
                dga[j] += dna * xa; dba[j] += dna; dgg[j] += dng * xg; dbg[j] += dng;
                const float ha = dna * ga[j], hg = dng * gg[j];
                s1a += ha; s2a += ha * xa; s1g += hg; s2g += hg * xg;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xa = (a.v[j] - mua) * rsa;
                const float na = xa * ga[j] + ba[j];
                dsc[j] += d.v[j] * gelu(na);
                const float dna = d.v[j] * sc[j] * gelu_grad(na);
                dga[j] += dna * xa; dba[j] += dna;
                const float ha = dna * ga[j];
                s1a += ha; s2a += ha * xa;
            }
        }
      }
    }
    // per-channel partials of the block: the lanes of a wave that hold the same channels (nq apart) meet by xor-shuffles, then the four
    // waves add their words one after the other with plain read-add-write -- ds_add_f32 from every lane at once costs ~200 cycles
    // per wave instruction when several lanes / waves hit a word (measured in csrc/tasnet.hip: 0.34 ms of ConvTasNet's 4.4-ms step)
    if ((nq & (nq - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            for (int o = nq; o < 64; o <<= 1) {
                dga[j] += __shfl_xor(dga[j], o, 64); dba[j] += __shfl_xor(dba[j], o, 64);
                if (MODE) { dgg[j] += __shfl_xor(dgg[j], o, 64); dbg[j] += __shfl_xor(dbg[j], o, 64); }
                if (scale) dsc[j] += __shfl_xor(dsc[j], o, 64);
            }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int turn = 0; turn < 4; ++turn) {
            if (wave == turn && lane < nq) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    lds[c0 + j] += dga[j];
                    lds[C + c0 + j] += dba[j];
                    if (MODE) { lds[c1 + j] += dgg[j]; lds[C + c1 + j] += dbg[j]; }
                    if (scale) lds[2 * C + c0 + j] += dsc[j];
                }
            }
            __syncthreads();
        }
    } else {
        // any other piece count: runs of nq consecutive threads hold distinct channels (piece index = global thread index mod nq); the
        // runs add one after the other -- a fixed order (it was LDS atomics: the hardware's order, different from run to run)
        const int run = threadIdx.x / nq, nruns = (256 + nq - 1) / nq;
        for (int turn = 0; turn < nruns; ++turn) {
            if (run == turn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    lds[c0 + j] += dga[j];
                    lds[C + c0 + j] += dba[j];
                    if (MODE) { lds[c1 + j] += dgg[j]; lds[C + c1 + j] += dbg[j]; }
                    if (scale) lds[2 * C + c0 + j] += dsc[j];
                }
            }
            __syncthreads();
        }
    }
    if (dc.part == nullptr) {
        atomicAdd(&lds[NV + 2 * grp_a], s1a);
        atomicAdd(&lds[NV + 2 * grp_a + 1], s2a);
        if (MODE) { atomicAdd(&lds[NV + 2 * grp_g], s1g); atomicAdd(&lds[NV + 2 * grp_g + 1], s2g); }
        __syncthreads();
    } else {         // deterministic schedule: the threads' group sums in thread order (csrc/det.h)
        __shared__ float tmpv[256];
        __shared__ int tmps[256];
        det_slot_sum(s1a, 2 * grp_a, 2 * G, lds + NV, tmpv, tmps, true);
        det_slot_sum(s2a, 2 * grp_a + 1, 2 * G, lds + NV, tmpv, tmps, true);
        if (MODE) {
            det_slot_sum(s1g, 2 * grp_g, 2 * G, lds + NV, tmpv, tmps, true);
            det_slot_sum(s2g, 2 * grp_g + 1, 2 * G, lds + NV, tmpv, tmps, true);
        }
    }
    if (part) {
        float* row = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NV;
        for (int i = threadIdx.x; i < NV; i += 256) row[i] = lds[i];
    } else {
        for (int i = threadIdx.x; i < NV; i += 256)
            if (lds[i] != 0.f) atomicAdd(&gch[i], lds[i]);
    }
    det_group_add(lds + NV, 2 * G, sums + (long)b * G * 2, dc, b, blockIdx.x, gridDim.x);
}

// gch[c] += sum_r part[r][c], spread over the workgroups of the launch that hosts it as units of (256 columns, one of <= 64 row
// groups): each unit adds its rows and flushes with one atomic per column (csrc/tasnet.hip ctn_gln_bwd_apply_kernel does the same)
__device__ __forceinline__ void dmx_colsum_share(const float* __restrict__ part, int nrows, int ncols, float* __restrict__ gch, int det) {
    const int nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int ncb = (ncols + 255) >> 8;
    int rg = nrows >> 3;
    rg = rg > 64 ? 64 : (rg < 1 ? 1 : rg);
    if (det) rg = 1;           // deterministic schedule: one unit per 256 columns adds ALL rows in row order, one add per column
    for (int u = bid; u < ncb * rg; u += nblk) {
        const int c = (u % ncb) * 256 + threadIdx.x, r0 = u / ncb;
        if (c < ncols) {
            float acc = 0.f;
            for (int r = r0; r < nrows; r += rg) acc += part[(size_t)r * ncols + c];
            if (acc != 0.f) atomicAdd(&gch[c], acc);
        }
    }
}

// pass 2: dy[b][t][C]
template <int MODE>
__global__ __launch_bounds__(256) void dmx_act_bwd_apply_kernel(const bf16_raw* __restrict__ dz, const bf16_raw* __restrict__ y,
                                                                const double* __restrict__ stats, const double* __restrict__ sums,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, int G, float eps,
                                                                const float* __restrict__ scale, int T, int C, bf16_raw* __restrict__ dy,
                                                                const float* __restrict__ part, int nrows, float* __restrict__ gch, int det) {
    if (part) dmx_colsum_share(part, nrows, 2 * C + (MODE ? C >> 1 : C), gch, det);      // the reduce pass's partial rows -> gch (see there)
    const int b = blockIdx.y;
    const int Co = MODE ? C >> 1 : C, nq = Co >> 3;
    const Walk w = walk(nq);
    const int c0 = w.q * 8, c1 = (C >> 1) + c0;
    const bool norm = stats != nullptr;
    float ga[8], ba[8], gg[8], bg[8], sc[8];
    float mua = 0.f, rsa = 1.f, mug = 0.f, rsg = 1.f, k1a = 0.f, k2a = 0.f, k1g = 0.f, k2g = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = gg[j] = 1.f; ba[j] = bg[j] = 0.f; sc[j] = 1.f; }
    if (norm) {
        const double n = (double)T * (C / G);
        const int grp_a = c0 / (C / G), grp_g = c1 / (C / G);
        ld8f(gamma + c0, ga); ld8f(beta + c0, ba);
        moments(stats, b, G, grp_a, n, eps, mua, rsa);
        k1a = (float)(sums[((long)b * G + grp_a) * 2] / n); k2a = (float)(sums[((long)b * G + grp_a) * 2 + 1] / n);
        if (MODE) {
            ld8f(gamma + c1, gg); ld8f(beta + c1, bg);
            moments(stats, b, G, grp_g, n, eps, mug, rsg);
            k1g = (float)(sums[((long)b * G + grp_g) * 2] / n); k2g = (float)(sums[((long)b * G + grp_g) * 2 + 1] / n);
        }
    }
    if (scale) ld8f(scale + c0, sc);
    const bf16_raw* yb = y + (long)b * T * C;
    bf16_raw* ob = dy + (long)b * T * C;
    for (int t0 = w.t0; t0 < T; t0 += DMX_FU * w.tstep) {       // DMX_FU frames per trip, their loads in flight together
        D8 a[DMX_FU], d[DMX_FU], g[DMX_FU];
        bool ok[DMX_FU];
#pragma unroll
        for (int u = 0; u < DMX_FU; ++u) {
            const int t = t0 + u * w.tstep;
            ok[u] = t < T;
            const int tt = ok[u] ? t : t0;
            a[u] = ld8(yb + (long)tt * C + c0);
            d[u] = ld8(dz + ((long)b * T + tt) * Co + c0);
            if (MODE) g[u] = ld8(yb + (long)tt * C + c1);
        }
#pragma unroll
        for (int u = 0; u < DMX_FU; ++u) {
            if (!ok[u]) continue;
            const int t = t0 + u * w.tstep;
            float oa[8], og[8];
            if (MODE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xa = (a[u].v[j] - mua) * rsa, xg = (g[u].v[j] - mug) * rsg;
                    const float na = xa * ga[j] + ba[j], sg = sigm(xg * gg[j] + bg[j]);
                    const float dv = d[u].v[j] * sc[j];
                    const float dna = dv * sg, dng = dv * na * sg * (1.f - sg);
                    oa[j] = norm ? (dna * ga[j] - k1a - xa * k2a) * rsa : dna;
                    og[j] = norm ? (dng * gg[j] - k1g - xg * k2g) * rsg : dng;
                }
                st8(ob + (long)t * C + c0, oa);
                st8(ob + (long)t * C + c1, og);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xa = (a[u].v[j] - mua) * rsa;
                    const float dna = d[u].v[j] * sc[j] * gelu_grad(xa * ga[j] + ba[j]);
                    oa[j] = norm ? (dna * ga[j] - k1a - xa * k2a) * rsa : dna;
                }
                st8(ob + (long)t * C + c0, oa);
            }
        }
    }
}

__global__ __launch_bounds__(256) void dmx_add_kernel(const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b, long n8,
                                                      bf16_raw* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const D8 x = ld8(a + i * 8), y = ld8(b + i * 8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x.v[j] + y.v[j];
        st8(out + i * 8, v);
    }
}

// BLSTM's overlapping chunks (src/model/demucs.py:91-117): W frames at hop S, nf = ceil(T / S) chunks per item (nf == 1: W == T, the
// whole sequence).  mode 0 gather : out[b nf + k][j] = a[b][S k + j] (0 beyond T)                                  (unfold :17-32)
//                   mode 1 pick   : out[b][t] = a[b nf + k(t)][t - S k(t)] + b[b][t], k(t) = the chunk whose middle part holds t (:104-117)
//                   mode 2 select : out[b nf + k][j] = (k == k(t)) ? a[b][t] : 0, t = S k + j                       (adjoint of pick)
//                   mode 3 sum    : out[b][t] = b[b][t] + sum_k a[b nf + k][t - S k]                               (adjoint of gather)
__device__ __forceinline__ int chunk_of(int t, int nf, int W, int S) {
    if (nf == 1 || t < W - S / 2) return 0;
    const int k = (t - S / 2) / S;
    return k < nf ? k : nf - 1;
}
__global__ __launch_bounds__(256) void dmx_frames_kernel(int mode, const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b2, int B, int T, int C,
                                                         int nf, int W, int S, bf16_raw* __restrict__ out) {
    const int nq = C >> 3;
    const bool framed_out = mode == 0 || mode == 2;
    const long total = (framed_out ? (long)B * nf * W : (long)B * T) * nq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % nq);
        const long r = i / nq;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (framed_out) {
            const int j = (int)(r % W), k = (int)((r / W) % nf), b = (int)(r / ((long)W * nf));
            const int t = S * k + j;
            if (t < T && (mode == 0 || chunk_of(t, nf, W, S) == k)) {
                const D8 x = ld8(a + ((long)b * T + t) * C + 8 * q);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x.v[e];
            }
        } else {
            const int t = (int)(r % T), b = (int)(r / T);
            const D8 x = ld8(b2 + r * C + 8 * q);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = x.v[e];
            if (mode == 1) {
                const int k = chunk_of(t, nf, W, S);
                const D8 y = ld8(a + (((long)b * nf + k) * W + (t - S * k)) * C + 8 * q);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += y.v[e];
            } else {
                for (int k = t / S; k >= 0 && k >= t / S - 1; --k) {
                    if (k >= nf || t - S * k >= W) continue;
                    const D8 y = ld8(a + (((long)b * nf + k) * W + (t - S * k)) * C + 8 * q);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += y.v[e];
                }
            }
        }
        st8(out + r * C + 8 * q, v);
    }
}

__global__ __launch_bounds__(256) void dmx_f32_to_bf16_kernel(const float* __restrict__ a, long n, bf16_raw* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = f2bf(a[i]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// bidirectional LSTM layer, one time step per launch.  grid (H/16, 2 directions, ceil(Bn/16)); 4 waves.
//   pre   fp32 [Bn][T][2][4][H]: x W_ih^T + b_ih + b_hh on entry, the ACTIVATED gates (i, f, g, o) on exit (kept for backward)
//   whh   bf16 [2][4H][H];  hs bf16 [Bn][T][2H] (output = next step's operand);  cs fp32 [Bn][T][2H]
// Forward: wave w computes gate w of 16 units for 16 batch rows with 16x16x32 MFMAs (A = h(t-1), B = W_hh rows), the gates meet
// in LDS, thread (row, unit) does the cell update.  Direction 1 walks the sequence backwards.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dmx_lstm_step_fwd_kernel(float* __restrict__ pre, const bf16_raw* __restrict__ whh,
                                                                bf16_raw* __restrict__ hs, float* __restrict__ cs, int Bn, int T, int H, int s) {
    __shared__ float gl[4][16][17];
    const int dir = blockIdx.y, u0 = blockIdx.x * 16, bt = blockIdx.z * 16;
    const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, ug = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        const int brow = bt + m;
        const bf16_raw* hp = hs + ((long)(brow < Bn ? brow : Bn - 1) * T + tp) * 2 * H + dir * H + 8 * ug;
        const bf16_raw* wp = whh + ((long)(dir * 4 + w) * H + u0 + m) * H + 8 * ug;
        for (int k0 = 0; k0 < H; k0 += 32) {
            uint4 av = *reinterpret_cast<const uint4*>(hp + k0);
            if (brow >= Bn) av = make_uint4(0, 0, 0, 0);
            const uint4 bv = *reinterpret_cast<const uint4*>(wp + k0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) gl[w][4 * ug + r][m] = acc[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, u = threadIdx.x & 15, b = bt + row;
    if (b >= Bn) return;
    float* pg = pre + (((long)b * T + t) * 2 + dir) * 4 * H + u0 + u;
    const float gi = sigm(pg[0] + gl[0][row][u]);
    const float gf = sigm(pg[H] + gl[1][row][u]);
    const float gg = tanhf(pg[2 * H] + gl[2][row][u]);
    const float go = sigm(pg[3 * H] + gl[3][row][u]);
    const long o = ((long)b * T + t) * 2 * H + dir * H + u0 + u;
    const float cp = s > 0 ? cs[((long)b * T + tp) * 2 * H + dir * H + u0 + u] : 0.f;
    const float c = gf * cp + gi * gg;
    cs[o] = c;
    hs[o] = f2bf(go * tanhf(c));
    pg[0] = gi; pg[H] = gf; pg[2 * H] = gg; pg[3 * H] = go;
}

// Backward step: dh = dhs[t] + dG(next) W_hh (wave w reduces the K quarter of gate w), then the cell backward.
//   whhT bf16 [2][H][4H];  dhs bf16 [Bn][T][2H] gradient of the layer output;  dG bf16 [Bn][T][2][4H] pre-activation gate gradients
//   dc fp32 [2][Bn][H] running cell-state gradient
__global__ __launch_bounds__(256) void dmx_lstm_step_bwd_kernel(const float* __restrict__ gates, const bf16_raw* __restrict__ whhT,
                                                                const float* __restrict__ cs, const bf16_raw* __restrict__ dhs,
                                                                bf16_raw* __restrict__ dG, float* __restrict__ dc, int Bn, int T, int H, int s) {
    __shared__ float pl[4][16][17];
    const int dir = blockIdx.y, u0 = blockIdx.x * 16, bt = blockIdx.z * 16;
    const int t = dir ? s : T - 1 - s;          // reverse of the forward order
    const int tn = dir ? t - 1 : t + 1;         // the step after t in forward order (processed by the previous launch)
    const int tp = dir ? t + 1 : t - 1;         // the step before t in forward order
    const bool has_prev = dir ? (t < T - 1) : (t > 0);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, ug = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (s > 0) {
        const int brow = bt + m;
        const bf16_raw* ap = dG + (((long)(brow < Bn ? brow : Bn - 1) * T + tn) * 2 + dir) * 4 * H + w * H + 8 * ug;
        const bf16_raw* wp = whhT + ((long)dir * H + u0 + m) * 4 * H + w * H + 8 * ug;
        for (int k0 = 0; k0 < H; k0 += 32) {
            uint4 av = *reinterpret_cast<const uint4*>(ap + k0);
            if (brow >= Bn) av = make_uint4(0, 0, 0, 0);
            const uint4 bv = *reinterpret_cast<const uint4*>(wp + k0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) pl[w][4 * ug + r][m] = acc[r];
    __syncthreads();
    const int row = threadIdx.x >> 4, u = threadIdx.x & 15, b = bt + row;
    if (b >= Bn) return;
    const long o = ((long)b * T + t) * 2 * H + dir * H + u0 + u;
    const float dh = bf2f(dhs[o]) + pl[0][row][u] + pl[1][row][u] + pl[2][row][u] + pl[3][row][u];
    const float* pg = gates + (((long)b * T + t) * 2 + dir) * 4 * H + u0 + u;
    const float gi = pg[0], gf = pg[H], gg = pg[2 * H], go = pg[3 * H];
    const float c = cs[o];
    const float cp = has_prev ? cs[((long)b * T + tp) * 2 * H + dir * H + u0 + u] : 0.f;
    float* dcp = dc + ((long)dir * Bn + b) * H + u0 + u;
    const float tc = tanhf(c);
    const float dcc = dh * go * (1.f - tc * tc) + (s > 0 ? *dcp : 0.f);
    *dcp = dcc * gf;
    bf16_raw* dg = dG + (((long)b * T + t) * 2 + dir) * 4 * H + u0 + u;
    dg[0] = f2bf(dcc * gg * gi * (1.f - gi));
    dg[H] = f2bf(dcc * cp * gf * (1.f - gf));
    dg[2 * H] = f2bf(dcc * gi * (1.f - gg * gg));
    dg[3 * H] = f2bf(dh * tc * go * (1.f - go));
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same layer as ONE persistent launch: the workgroups of a (direction, batch tile) group -- H/16 of them, each owning 16 hidden
// units with its W_hh slice resident in registers -- hand h(t) (forward) / the gate gradients (backward) to each other through
// global memory once per time step.  Hand-off protocol (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the table of
// sc1 hand-offs): every handed-off byte is stored with an 8-byte agent-scope relaxed atomic store (sc1: write-through), every
// storing wave drains vmcnt, a workgroup barrier, then the group's arrival counter is advanced (agent-scope atomics; the counter is
// kept in 32 replicas on cache lines of their own, one wave instruction adds to all of them and a member polls replica member % 32:
// 32 pollers on ONE line cost 0.8 ms per Demucs step: 23.6 ms with one counter, 23.1 with 8 replicas, 22.8 with 32); the consumer's lane 0 polls its replica with sc1 loads (bounded spin: a time-out sets sync[TMO] and lets the kernel run to its end with
// garbage instead of hanging; the word is sticky: the call clears the counters only, the owner of the block reads it when it
// likes), a workgroup barrier, then EVERY load of handed-off bytes is an sc1 load to registers.
// No fence, no L2 write-back.  Results never depend on placement; for speed the members of a group are given equal
// workgroup-id % 8 (one XCD under round-robin dispatch).  The host wrapper zeroes the sync block before every launch and falls
// back to the per-step launches when the active workgroups could not all be resident (> 256) or H/32 is not 1, 2, 4, 8 or 16.
// ---------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define DMX_SYNC_WORDS 16384       // sync block: word 60 = time-out (sticky); from word 64: per group 32 replicas of its arrival counter, 128 B apart
#define DMX_TMO 60
#define DMX_LIM 61                 // test hook: a non-zero word here replaces DMX_SPIN_LIMIT; 0xffffffff = time out at the first wait (tests/test_gpu_demucs.py)
#define DMX_CNT0 64
#define DMX_REPL 32
#define DMX_RSTRIDE 32
#define DMX_SPIN_LIMIT (1u << 20)  // polls of >= 0.3 us each: a fraction of a second, once (the time-out is sticky)

struct SeqMap { int dir, bz, member; bool active; };
__device__ __forceinline__ SeqMap seq_map(int nb, int btiles) {
    const int id = blockIdx.x, lab = id & 7, slot = id >> 3;
    const int group = lab + 8 * (slot / nb);
    SeqMap m;
    m.member = slot % nb;
    m.active = group < 2 * btiles;
    m.dir = group & 1;
    m.bz = group >> 1;
    return m;
}
// lane 0 of wave 0 waits until the group's counter has reached `target`, then the workgroup barrier
__device__ __forceinline__ void group_wait(gu32* cnt, unsigned target, gu32* tmo, bool& dead, unsigned limit) {
    if (threadIdx.x == 0 && !dead && limit == 0xffffffffu) {      // test hook: the first wait of the launch "times out" whatever the timing
        __hip_atomic_store(tmo, 1u, RLX_AGENT);
        dead = true;
    }
    if (threadIdx.x == 0 && !dead) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, RLX_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > limit || ((spins & 1023u) == 0 && __hip_atomic_load(tmo, RLX_AGENT) != 0)) {
                __hip_atomic_store(tmo, 1u, RLX_AGENT);
                dead = true;      // give up for the rest of the sequence: garbage out, but the launch ends
                break;
            }
        }
    }
    __syncthreads();
}
// N 16-byte sc1 loads (buffer_load_dwordx4 ... sc1: L1 bypassed, tracked by the compiler's vmcnt bookkeeping) of the elements
// base[off + stride * k ..+7]; rsrc describes the whole tensor (wave-uniform), offsets are per lane.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int N>
__device__ __forceinline__ void ld16_sc1_n(__amdgpu_buffer_rsrc_t rsrc, long off_elems, int stride_elems, uint4 (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((off_elems + (long)stride_elems * k) * 2), 0, 16);
        v[k] = make_uint4(x[0], x[1], x[2], x[3]);
    }
}

template <int KS>
__global__ __launch_bounds__(256) void dmx_lstm_seq_fwd_kernel(float* __restrict__ pre, const bf16_raw* __restrict__ whh, bf16_raw* hs,
                                                               float* __restrict__ cs, int Bn, int T, int btiles, unsigned* sync) {
    constexpr int H = 32 * KS;
    __shared__ float gl[4][16][17];
    __shared__ __attribute__((aligned(16))) bf16_raw hrow[16][16];
    __shared__ uint4 afr[KS >= 4 ? KS : 1][64];
    const SeqMap sm = seq_map(H / 16, btiles);
    if (!sm.active) return;
    const int dir = sm.dir, u0 = sm.member * 16, bt = sm.bz * 16, nb = H / 16;
    gu32* cnt0 = (gu32*)sync + DMX_CNT0 + (sm.bz * 2 + dir) * DMX_REPL * DMX_RSTRIDE;      // every member adds to all replicas,
    gu32* cnt = cnt0 + (sm.member % DMX_REPL) * DMX_RSTRIDE;                                    // and polls its own
    gu32* tmo = (gu32*)sync + DMX_TMO;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, ug = lane >> 4;
    bf16x8 wf[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k)
        wf[k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(whh + ((long)(dir * 4 + w) * H + u0 + m) * H + 32 * k + 8 * ug));
    const int row = threadIdx.x >> 4, u = threadIdx.x & 15, b = bt + row;
    const bool bok = b < Bn;
    const int brow = bt + m < Bn ? bt + m : Bn - 1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(hs, 0, (int)((long)Bn * T * 2 * H * 2), 0x00020000);
    float c = 0.f;
    bool dead = false;
    const unsigned spin_limit = sync[DMX_LIM] ? sync[DMX_LIM] : DMX_SPIN_LIMIT;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s, tp = dir ? t + 1 : t - 1;
        float* pg = pre + (((long)(bok ? b : Bn - 1) * T + t) * 2 + dir) * 4 * H + u0 + u;
        const float p0 = pg[0], p1 = pg[H], p2 = pg[2 * H], p3 = pg[3 * H];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            group_wait(cnt, (unsigned)(nb * s), tmo, dead, spin_limit);
            const long hp = ((long)brow * T + tp) * 2 * H + dir * H + 8 * ug;
            uint4 av[KS];
            if constexpr (KS >= 4) {
                // the four gate waves need the same h fragments: each fetches a quarter, they meet in LDS
                uint4 part[KS / 4];
                ld16_sc1_n<KS / 4>(rsrc, hp + 32 * (KS / 4) * w, 32, part);
#pragma unroll
                for (int k = 0; k < KS / 4; ++k) afr[(KS / 4) * w + k][lane] = part[k];
                __syncthreads();
#pragma unroll
                for (int k = 0; k < KS; ++k) av[k] = afr[k][lane];
            } else {
                ld16_sc1_n<KS>(rsrc, hp, 32, av);
            }
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (bt + m >= Bn) av[k] = make_uint4(0, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[k]), wf[k], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) gl[w][4 * ug + r][m] = acc[r];
        __syncthreads();
        const float gi = sigm(p0 + gl[0][row][u]), gf = sigm(p1 + gl[1][row][u]), gg = tanhf(p2 + gl[2][row][u]), go = sigm(p3 + gl[3][row][u]);
        c = gf * c + gi * gg;
        const bf16_raw hb = f2bf(go * tanhf(c));
        hrow[row][u] = hb;
        if (bok) {
            cs[((long)b * T + t) * 2 * H + dir * H + u0 + u] = c;
            pg[0] = gi; pg[H] = gf; pg[2 * H] = gg; pg[3 * H] = go;
        }
        __syncthreads();
        if (w == 0) {       // publish the 16 x 16 tile: one 8-byte write-through store per lane, drain, signal
            const int r = lane >> 2, part = lane & 3;
            if (bt + r < Bn)
                __hip_atomic_store((gu64*)(hs + ((long)(bt + r) * T + t) * 2 * H + dir * H + u0 + 4 * part),
                                   *reinterpret_cast<const unsigned long long*>(&hrow[r][4 * part]), RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < DMX_REPL) __hip_atomic_fetch_add(cnt0 + lane * DMX_RSTRIDE, 1u, RLX_AGENT);
        }
    }
}

template <int KS>
__global__ __launch_bounds__(256) void dmx_lstm_seq_bwd_kernel(const float* __restrict__ gates, const bf16_raw* __restrict__ whhT,
                                                               const float* __restrict__ cs, const bf16_raw* __restrict__ dhs, bf16_raw* dG,
                                                               int Bn, int T, int btiles, unsigned* sync) {
    constexpr int H = 32 * KS;
    __shared__ float pl[4][16][17];
    __shared__ __attribute__((aligned(16))) bf16_raw dgt[4][16][16];
    const SeqMap sm = seq_map(H / 16, btiles);
    if (!sm.active) return;
    const int dir = sm.dir, u0 = sm.member * 16, bt = sm.bz * 16, nb = H / 16;
    gu32* cnt0 = (gu32*)sync + DMX_CNT0 + (sm.bz * 2 + dir) * DMX_REPL * DMX_RSTRIDE;      // every member adds to all replicas,
    gu32* cnt = cnt0 + (sm.member % DMX_REPL) * DMX_RSTRIDE;                                    // and polls its own
    gu32* tmo = (gu32*)sync + DMX_TMO;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, ug = lane >> 4;
    bf16x8 wf[KS];      // W_hh^T rows u0 + m, K quarter of gate w
#pragma unroll
    for (int k = 0; k < KS; ++k)
        wf[k] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(whhT + ((long)dir * H + u0 + m) * 4 * H + w * H + 32 * k + 8 * ug));
    const int row = threadIdx.x >> 4, u = threadIdx.x & 15, b = bt + row;
    const bool bok = b < Bn;
    const int bc = bok ? b : Bn - 1;
    const int brow = bt + m < Bn ? bt + m : Bn - 1;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dG, 0, (int)((long)Bn * T * 8 * H * 2), 0x00020000);
    float dcs = 0.f;
    bool dead = false;
    const unsigned spin_limit = sync[DMX_LIM] ? sync[DMX_LIM] : DMX_SPIN_LIMIT;
    for (int s = 0; s < T; ++s) {
        const int t = dir ? s : T - 1 - s, tn = dir ? t - 1 : t + 1, tp = dir ? t + 1 : t - 1;
        const bool has_prev = dir ? (t < T - 1) : (t > 0);
        const long o = ((long)bc * T + t) * 2 * H + dir * H + u0 + u;
        const float* pg = gates + (((long)bc * T + t) * 2 + dir) * 4 * H + u0 + u;
        const float gi = pg[0], gf = pg[H], gg = pg[2 * H], go = pg[3 * H];
        const float c = cs[o], cp = has_prev ? cs[((long)bc * T + tp) * 2 * H + dir * H + u0 + u] : 0.f;
        const float dh_up = bf2f(dhs[o]);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            group_wait(cnt, (unsigned)(nb * s), tmo, dead, spin_limit);
            const long ap = (((long)brow * T + tn) * 2 + dir) * 4 * H + w * H + 8 * ug;
            uint4 av[KS];
            ld16_sc1_n<KS>(rsrc, ap, 32, av);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                if (bt + m >= Bn) av[k] = make_uint4(0, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[k]), wf[k], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pl[w][4 * ug + r][m] = acc[r];
        __syncthreads();
        const float dh = dh_up + pl[0][row][u] + pl[1][row][u] + pl[2][row][u] + pl[3][row][u];
        const float tc = tanhf(c);
        const float dcc = dh * go * (1.f - tc * tc) + dcs;
        dcs = dcc * gf;
        dgt[0][row][u] = f2bf(dcc * gg * gi * (1.f - gi));
        dgt[1][row][u] = f2bf(dcc * cp * gf * (1.f - gf));
        dgt[2][row][u] = f2bf(dcc * gi * (1.f - gg * gg));
        dgt[3][row][u] = f2bf(dh * tc * go * (1.f - go));
        __syncthreads();
        {   // wave w publishes the tile of gate w: one 8-byte write-through store per lane; every wave drains; barrier; one lane signals
            const int r = lane >> 2, part = lane & 3;
            if (bt + r < Bn)
                __hip_atomic_store((gu64*)(dG + (((long)(bt + r) * T + t) * 2 + dir) * 4 * H + w * H + u0 + 4 * part),
                                   *reinterpret_cast<const unsigned long long*>(&dgt[w][r][4 * part]), RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x < DMX_REPL) __hip_atomic_fetch_add(cnt0 + threadIdx.x * DMX_RSTRIDE, 1u, RLX_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LocalState attention.  qkv bf16 [B][T][NQ]: query [0, hid) | key [hid, 2 hid) | content [2 hid, 3 hid) | decay [3 hid, 3 hid + heads nd).
// One workgroup = one (batch, head, tile of QT queries) and ALL keys; scores live in LDS as sc[t][QT].
// ---------------------------------------------------------------------------------------------------------------------------
#define QT 32

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// dot product of two rows of c bf16 elements, fp32 accumulation: v_dot2c_f32_bf16 takes the pairs as they are stored (4 instructions
// per 8 elements instead of 16 unpacking + 8 multiply-adds: these loops are VALU-bound)
template <int NCH>     // NCH = c / 8 known at compile time (1 .. 16): all row chunks are requested before the first use
__device__ __forceinline__ float dot_chunks(const bf16_raw* a, const bf16_raw* b, int c) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < (NCH ? 8 * NCH : c); j += 8) {
        const uint4 x = *reinterpret_cast<const uint4*>(a + j), y = *reinterpret_cast<const uint4*>(b + j);
        s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.x), __builtin_bit_cast(bf16x2, y.x), s0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.y), __builtin_bit_cast(bf16x2, y.y), s1, false);
        s0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.z), __builtin_bit_cast(bf16x2, y.z), s0, false);
        s1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x.w), __builtin_bit_cast(bf16x2, y.w), s1, false);
    }
    return s0 + s1;
}
// acc[0..8) += w * row[0..8) on packed pairs (v_pk_fma_f32: 4 multiply-add instructions per 8 elements)
struct Acc8 { f32x2 v[4]; };
__device__ __forceinline__ void axpy8(Acc8& acc, float w, const bf16_raw* row) {
    const uint4 u = *reinterpret_cast<const uint4*>(row);
    const unsigned d[4] = {u.x, u.y, u.z, u.w};
    const f32x2 ww = {w, w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 x = {__uint_as_float(d[i] << 16), __uint_as_float(d[i] & 0xffff0000u)};
        acc.v[i] = __builtin_elementwise_fma(ww, x, acc.v[i]);
    }
}
__device__ __forceinline__ Acc8 acc8_zero() {
    Acc8 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a.v[i] = (f32x2){0.f, 0.f};
    return a;
}

// Rows of one head as the kernels read them: row r at p + r * stride.  Either straight from the tensors or, where the LDS has
// room beside the score tiles (`stage`), from a copy made once per workgroup (pitch c + 8 elements: a wave's 32 query rows then
// start in different banks): every key / content row is read by all 32 queries and every query row by all T keys, and from
// memory that was 340 000 sixteen-byte loads per workgroup (430 us per launch at T = 187).
struct Rows { const bf16_raw* p; int stride; };
__device__ __forceinline__ Rows stage_rows(const bf16_raw* __restrict__ src, int src_stride, int nrows, int nvalid, int c, bf16_raw* lds, bool stage) {
    if (!stage) return Rows{src, src_stride};
    const int nch = c >> 3, pitch = c + 8;
    for (int i = threadIdx.x; i < nrows * nch; i += 256) {
        const int r = i / nch, j = i % nch;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r < nvalid) v = *reinterpret_cast<const uint4*>(src + (long)r * src_stride + 8 * j);
        *reinterpret_cast<uint4*>(lds + r * pitch + 8 * j) = v;
    }
    return Rows{lds, pitch};
}

// scores -> softmax weights in sc (columns of queries beyond T hold zeros); dsum[sl] = sum_f (f+1) sigmoid(raw_f)/2 / sqrt(nd)
// K: key rows 0..T; Q: the tile's query rows 0..QT; raw: decay logits of query s0 at raw + 0, row stride NQ
template <int NCH>
__device__ __forceinline__ void attn_weights(Rows K, Rows Q, const bf16_raw* __restrict__ raw, int NQ, int T, int c, int nd, int s0,
                                             float* __restrict__ sc, float* __restrict__ dsum, float* __restrict__ red) {
    const int tid = threadIdx.x;
    if (tid < QT) {
        const int s = s0 + tid;
        float v = 0.f;
        if (s < T)
            for (int f = 0; f < nd; ++f) v += (float)(f + 1) * 0.5f * sigm(bf2f(raw[(long)tid * NQ + f]));
        dsum[tid] = v * rsqrtf((float)nd);
    }
    __syncthreads();
    const float isq = rsqrtf((float)c);
    for (int p = tid; p < T * QT; p += 256) {
        const int t = p / QT, sl = p % QT, s = s0 + sl;
        float v = 0.f;
        if (s < T) {
            v = dot_chunks<NCH>(K.p + (long)t * K.stride, Q.p + (long)sl * Q.stride, c) * isq - fabsf((float)(t - s)) * dsum[sl];
            if (t == s) v = -100.f;
        }
        sc[p] = v;
    }
    __syncthreads();
    // softmax over t per column: 8 threads per column
    const int sl = tid % QT, sub = tid / QT;
    float mx = -3.0e38f;
    for (int t = sub; t < T; t += 8) mx = fmaxf(mx, sc[t * QT + sl]);
    red[sub * QT + sl] = mx;
    __syncthreads();
    mx = red[sl];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, red[i * QT + sl]);
    __syncthreads();
    float sum = 0.f;
    for (int t = sub; t < T; t += 8) { const float e = __expf(sc[t * QT + sl] - mx); sc[t * QT + sl] = e; sum += e; }
    red[sub * QT + sl] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += red[i * QT + sl];
    const float inv = (s0 + sl < T) ? 1.f / sum : 0.f;
    for (int t = sub; t < T; t += 8) sc[t * QT + sl] *= inv;
    __syncthreads();
}

template <int NCH>
__global__ __launch_bounds__(256) void dmx_attn_fwd_kernel(const bf16_raw* __restrict__ qkv, int T, int hid, int heads, int nd, int NQ, int stage,
                                                           bf16_raw* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                 // [T][QT]
    float* dsum = sc + (size_t)T * QT;   // [QT]
    float* red = dsum + QT;           // [8][QT]
    const int b = blockIdx.z, h = blockIdx.y, s0 = blockIdx.x * QT, c = hid / heads;
    bf16_raw* tiles = reinterpret_cast<bf16_raw*>(red + 8 * QT);      // staged rows: keys [T] | content [T] | queries [QT], pitch c + 8
    const bf16_raw* qb = qkv + (long)b * T * NQ;
    const int nq = T - s0 < QT ? T - s0 : QT;
    const Rows K = stage_rows(qb + hid + h * c, NQ, T, T, c, tiles, stage);
    const Rows Cn = stage_rows(qb + 2 * hid + h * c, NQ, T, T, c, tiles + (size_t)T * (c + 8), stage);
    const Rows Q = stage_rows(qb + (long)s0 * NQ + h * c, NQ, QT, nq, c, tiles + (size_t)2 * T * (c + 8), stage);
    if (stage) __syncthreads();
    attn_weights<NCH>(K, Q, qb + (long)s0 * NQ + 3 * hid + h * nd, NQ, T, c, nd, s0, sc, dsum, red);
    const int nch = c >> 3;
    for (int it = threadIdx.x; it < QT * nch; it += 256) {
        const int sl = it % QT, j = it / QT, s = s0 + sl;
        if (s >= T) continue;
        Acc8 a8 = acc8_zero();
#pragma unroll 8
        for (int t = 0; t < T; ++t) axpy8(a8, sc[t * QT + sl], Cn.p + (long)t * Cn.stride + 8 * j);
        const float acc[8] = {a8.v[0][0], a8.v[0][1], a8.v[1][0], a8.v[1][1], a8.v[2][0], a8.v[2][1], a8.v[3][0], a8.v[3][1]};
        st8(out + ((long)b * T + s) * hid + h * c + 8 * j, acc);
    }
}

// dqkv bf16 [B][T][NQ]: the query and decay columns get their final gradients here (a query tile belongs to one workgroup); the
// key / content gradients of this tile's queries go to slab[tile][b][t][key hid | content hid] fp32 (plain stores, every element
// written exactly once) and dmx_attn_bwd_sum_kernel adds the tiles up.  (fp32 atomics onto [B][T][2 hid] instead: 170 of 325 us.)
template <int NCH>
__global__ __launch_bounds__(256) void dmx_attn_bwd_kernel(const bf16_raw* __restrict__ qkv, const bf16_raw* __restrict__ dres, int T, int hid,
                                                           int heads, int nd, int NQ, int stage, float* __restrict__ slab,
                                                           bf16_raw* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                     // [T][QT] softmax weights
    float* dw = sc + (size_t)T * QT;      // [T][QT] d weights -> d scores
    float* dsum = dw + (size_t)T * QT;    // [QT]
    float* red = dsum + QT;               // [8][QT]
    const int b = blockIdx.z, h = blockIdx.y, s0 = blockIdx.x * QT, c = hid / heads;
    const int tid = threadIdx.x;
    bf16_raw* tiles = reinterpret_cast<bf16_raw*>(red + 8 * QT);      // keys [T] | content [T] | queries [QT] | d result [QT]
    const bf16_raw* qb = qkv + (long)b * T * NQ;
    const bf16_raw* db = dres + (long)b * T * hid + h * c;
    bf16_raw* gq = dqkv + (long)b * T * NQ;
    float* sl_out = slab + (((long)blockIdx.x * gridDim.z + b) * T) * 2 * hid;      // [t][key hid | content hid] of this tile
    const int nq = T - s0 < QT ? T - s0 : QT;
    const size_t tp = (size_t)(c + 8);
    const Rows K = stage_rows(qb + hid + h * c, NQ, T, T, c, tiles, stage & 1);
    const Rows Cn = stage_rows(qb + 2 * hid + h * c, NQ, T, T, c, tiles + T * tp, stage & 1);
    const Rows Q = stage_rows(qb + (long)s0 * NQ + h * c, NQ, QT, nq, c, tiles + 2 * T * tp, stage & 1);
    const Rows D = stage_rows(db + (long)s0 * hid, hid, QT, nq, c, tiles + (2 * T + QT) * tp, stage & 1);
    if (stage & 1) __syncthreads();
    attn_weights<NCH>(K, Q, qb + (long)s0 * NQ + 3 * hid + h * nd, NQ, T, c, nd, s0, sc, dsum, red);
    for (int p = tid; p < T * QT; p += 256) {
        const int t = p / QT, sl = p % QT, s = s0 + sl;
        dw[p] = s < T ? dot_chunks<NCH>(D.p + (long)sl * D.stride, Cn.p + (long)t * Cn.stride, c) : 0.f;
    }
    __syncthreads();
    const int sl = tid % QT, sub = tid / QT;
    float part = 0.f;
    for (int t = sub; t < T; t += 8) part += sc[t * QT + sl] * dw[t * QT + sl];
    red[sub * QT + sl] = part;
    __syncthreads();
    float dotw = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) dotw += red[i * QT + sl];
    for (int t = sub; t < T; t += 8) {
        const float v = sc[t * QT + sl] * (dw[t * QT + sl] - dotw);
        dw[t * QT + sl] = (t == s0 + sl) ? 0.f : v;      // masked_fill: no gradient through the diagonal
    }
    __syncthreads();
    const float isq = rsqrtf((float)c);
    const int nch = c >> 3;
    // keys and content: item (t, chunk j)
    for (int it = tid; it < T * nch; it += 256) {
        const int t = it / nch, j = it % nch;
        Acc8 ak = acc8_zero(), ac = acc8_zero();
#pragma unroll 4
        for (int q = 0; q < nq; ++q) {
            axpy8(ak, dw[t * QT + q] * isq, Q.p + (long)q * Q.stride + 8 * j);
            axpy8(ac, sc[t * QT + q], D.p + (long)q * D.stride + 8 * j);
        }
        float* gk = sl_out + (long)t * 2 * hid + h * c + 8 * j;
        float* gc = gk + hid;
        *reinterpret_cast<float4*>(gk) = make_float4(ak.v[0][0], ak.v[0][1], ak.v[1][0], ak.v[1][1]);
        *reinterpret_cast<float4*>(gk + 4) = make_float4(ak.v[2][0], ak.v[2][1], ak.v[3][0], ak.v[3][1]);
        *reinterpret_cast<float4*>(gc) = make_float4(ac.v[0][0], ac.v[0][1], ac.v[1][0], ac.v[1][1]);
        *reinterpret_cast<float4*>(gc + 4) = make_float4(ac.v[2][0], ac.v[2][1], ac.v[3][0], ac.v[3][1]);
    }
    // queries: item (sl, chunk j), owned by this workgroup alone
    for (int it = tid; it < QT * nch; it += 256) {
        const int q = it % QT, j = it / QT, s = s0 + q;
        if (s >= T) continue;
        Acc8 aq = acc8_zero();
#pragma unroll 8
        for (int t = 0; t < T; ++t) axpy8(aq, dw[t * QT + q] * isq, K.p + (long)t * K.stride + 8 * j);
        const float o[8] = {aq.v[0][0], aq.v[0][1], aq.v[1][0], aq.v[1][1], aq.v[2][0], aq.v[2][1], aq.v[3][0], aq.v[3][1]};
        st8(gq + (long)s * NQ + h * c + 8 * j, o);
    }
    // decay: d raw_f[s] = -(f+1)/sqrt(nd) * (sum_t dscore[t][s] |t - s|) * sigmoid'(raw_f) / 2
    if (tid < QT && s0 + tid < T) {
        const int s = s0 + tid;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += dw[t * QT + tid] * fabsf((float)(t - s));
        a *= -rsqrtf((float)nd);
        for (int f = 0; f < nd; ++f) {
            const float sg = sigm(bf2f(qb[(long)s * NQ + 3 * hid + h * nd + f]));
            gq[(long)s * NQ + 3 * hid + h * nd + f] = f2bf(a * (float)(f + 1) * 0.5f * sg * (1.f - sg));
        }
    }
}

// dqkv[b][t][hid .. 3 hid) = sum over the query tiles of slab[tile][b][t][0 .. 2 hid)
__global__ __launch_bounds__(256) void dmx_attn_bwd_sum_kernel(const float* __restrict__ slab, int ntiles, long rows /*B T*/, int hid, int NQ,
                                                               bf16_raw* __restrict__ dqkv) {
    const int nq = (2 * hid) >> 3;
    const long total = rows * nq;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / nq;
        const int q = (int)(i % nq);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ntiles; ++k) {
            const float* p = slab + ((long)k * rows + r) * 2 * hid + 8 * q;
            const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        st8(dqkv + r * NQ + hid + 8 * q, v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------
static int dmx_check(const char* who, int B, int T, int C, int mode, int G) {
    SEHIP_REQUIRE(B > 0 && T > 0, "%s: empty input", who);
    SEHIP_REQUIRE(C >= 8 && (C & 7) == 0, "%s: C=%d must be a multiple of 8", who, C);
    SEHIP_REQUIRE(!mode || (C & 15) == 0, "%s: GLU needs C=%d to be a multiple of 16", who, C);
    SEHIP_REQUIRE(G >= 1 && G <= 8 && C % G == 0 && ((C / G) & 7) == 0, "%s: %d groups over C=%d: groups must hold a multiple of 8 channels", who, G, C);
    return 0;
}
static int gcd_(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
// grid.x such that grid.x * 256 is a multiple of nq pieces, about 8 pieces per thread, at most ~2048 workgroups in all
static dim3 dmx_grid(int B, int T, int nq) {
    const int unit = nq / gcd_(nq, 256);
    long want = ((long)T * nq + 256 * 8 - 1) / (256 * 8);
    const long cap = 2048 / B > 1 ? 2048 / B : 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    const long gx = (want + unit - 1) / unit * unit;
    return dim3((unsigned)gx, (unsigned)B);
}

extern "C" int sehip_dmx_prep(const float* mix, int B, int ac, int acp, int T, int padl, int Tv, int normalize, int up, const float* kup,
                              int width, int KL, double* acc, float* ms, void* x_bf16, void* stream) {
    SEHIP_REQUIRE(B > 0 && ac > 0 && acp >= ac && T > 0 && Tv >= T + padl, "dmx_prep: bad sizes (B=%d ac=%d acp=%d T=%d padl=%d Tv=%d)", B, ac, acp, T, padl, Tv);
    SEHIP_REQUIRE(!up || (kup && KL > 0), "dmx_prep: up-sampling needs its kernels");
    if (normalize) {
        SEHIP_REQUIRE(acc, "dmx_prep: normalize needs the accumulator scratch (2 doubles per item)");
        SEHIP_REQUIRE(hipMemsetAsync(acc, 0, (size_t)B * 2 * sizeof(double), (hipStream_t)stream) == hipSuccess, "dmx_prep: clearing the accumulators failed");
        int gx = (T + 256 * 16 - 1) / (256 * 16);
        if (gx > 64) gx = 64;
        bool ok;
        const DetCtx dc = sehip_det_ctx((hipStream_t)stream, (size_t)gx * B * 2, &ok);
        if (!ok) return -2;
        dmx_moments_kernel<<<dim3(gx, B), 256, 0, (hipStream_t)stream>>>(mix, ac, T, acc, dc);
        if (int e = sehip_det_finish((hipStream_t)stream, dc, B, gx, 2, acc, 2)) return e;
    }
    dmx_moments_finish_kernel<<<(B + 63) / 64, 64, 0, (hipStream_t)stream>>>(acc, B, T, normalize, ms);
    const long T0 = up ? 2L * Tv : Tv;
    static const bool no_tiled = getenv("SEHIP_DMX_NO_TILED_RESAMPLE") != nullptr;
    if (up && !no_tiled && ac <= 8 && (size_t)ac * (128 + KL) * 4 <= 60 * 1024)
        dmx_prep_up_kernel<<<dim3((unsigned)((T0 + 255) / 256), B), 256, (size_t)ac * (128 + KL) * 4, (hipStream_t)stream>>>(
            mix, ms, kup, ac, acp, T, padl, Tv, width, KL, (bf16_raw*)x_bf16);
    else
    dmx_prep_kernel<<<dim3((unsigned)((T0 + 255) / 256), B), 256, 0, (hipStream_t)stream>>>(mix, ms, kup, ac, acp, T, padl, Tv, up, width, KL,
                                                                                           (bf16_raw*)x_bf16);
    SEHIP_CHECK_LAUNCH("dmx_prep");
    return 0;
}

extern "C" int sehip_dmx_post(const float* y, const float* ms, int B, int co, int cop, long Tf, int padl, int T, int down, const float* kdn,
                              int width, int KL, float* out, void* stream) {
    SEHIP_REQUIRE(B > 0 && co > 0 && cop >= co && T > 0 && Tf > 0, "dmx_post: bad sizes");
    SEHIP_REQUIRE((down ? Tf / 2 : Tf) >= padl + T, "dmx_post: the network output (%ld) is shorter than pad + clip (%d + %d)", Tf, padl, T);
    static const bool no_tiled = getenv("SEHIP_DMX_NO_TILED_RESAMPLE") != nullptr;
    if (down && !no_tiled && (size_t)co * (512 + KL) * 4 <= 60 * 1024)
        dmx_post_down_kernel<<<dim3((unsigned)((T + 255) / 256), B), 256, (size_t)co * (512 + KL) * 4, (hipStream_t)stream>>>(
            y, ms, kdn, co, cop, Tf, padl, T, width, KL, out);
    else
    dmx_post_kernel<<<dim3((unsigned)((T + 255) / 256), B), 256, 0, (hipStream_t)stream>>>(y, ms, kdn, co, cop, Tf, padl, T, down, width, KL, out);
    SEHIP_CHECK_LAUNCH("dmx_post");
    return 0;
}

extern "C" int sehip_dmx_post_bwd(const float* dout, const float* ms, int B, int co, int cop, long Tf, int padl, int T, int down,
                                  const float* kdn, int width, int KL, void* dy_bf16, void* stream) {
    SEHIP_REQUIRE(B > 0 && co > 0 && cop >= co && T > 0 && Tf > 0, "dmx_post_bwd: bad sizes");
    static const bool no_tiled = getenv("SEHIP_DMX_NO_TILED_RESAMPLE") != nullptr;
    if (down && !no_tiled && (size_t)co * (128 + KL / 2 + 4) * 4 <= 60 * 1024)
        dmx_post_bwd_down_kernel<<<dim3((unsigned)((Tf + 255) / 256), B), 256, (size_t)co * (128 + KL / 2 + 4) * 4, (hipStream_t)stream>>>(
            dout, ms, kdn, co, cop, Tf, padl, T, width, KL, (bf16_raw*)dy_bf16);
    else
    dmx_post_bwd_kernel<<<dim3((unsigned)((Tf + 255) / 256), B), 256, 0, (hipStream_t)stream>>>(dout, ms, kdn, co, cop, Tf, padl, T, down, width, KL,
                                                                                              (bf16_raw*)dy_bf16);
    SEHIP_CHECK_LAUNCH("dmx_post_bwd");
    return 0;
}

extern "C" int sehip_dmx_gn_stats(const void* y, int B, int T, int C, int G, double* stats, void* stream) {
    if (int e = dmx_check("dmx_gn_stats", B, T, C, 0, G)) return e;
    const dim3 grid = dmx_grid(B, T, C >> 3);
    bool ok;
    const DetCtx dc = sehip_det_ctx((hipStream_t)stream, (size_t)grid.x * B * 2 * G, &ok);
    if (!ok) return -2;
    dmx_gn_stats_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, T, C, G, stats, dc);
    if (int e = sehip_det_finish((hipStream_t)stream, dc, B, (int)grid.x, 2 * G, stats, 2L * G)) return e;
    SEHIP_CHECK_LAUNCH("dmx_gn_stats");
    return 0;
}

extern "C" int sehip_dmx_act_fwd(const void* y, const double* stats, const float* gamma, const float* beta, int G, float eps, int mode,
                                 const float* scale, const void* resid, const void* add, int B, int T, int C, void* out, void* stream) {
    if (int e = dmx_check("dmx_act_fwd", B, T, C, mode, stats ? G : 1)) return e;
    SEHIP_REQUIRE(!stats || (gamma && beta), "dmx_act_fwd: GroupNorm needs its affine terms");
    SEHIP_REQUIRE(!scale || resid, "dmx_act_fwd: LayerScale comes with the residual input");
    const int nq = (mode ? C >> 1 : C) >> 3;
    const dim3 g = dmx_grid(B, T, nq);
    if (mode)
        dmx_act_fwd_kernel<1><<<g, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, stats, gamma, beta, G, eps, scale, (const bf16_raw*)resid,
                                                                   (const bf16_raw*)add, T, C, (bf16_raw*)out);
    else
        dmx_act_fwd_kernel<0><<<g, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)y, stats, gamma, beta, G, eps, scale, (const bf16_raw*)resid,
                                                                   (const bf16_raw*)add, T, C, (bf16_raw*)out);
    SEHIP_CHECK_LAUNCH("dmx_act_fwd");
    return 0;
}

extern "C" int sehip_dmx_act_bwd(const void* dz, const void* y, const double* stats, const float* gamma, const float* beta, int G, float eps,
                                 int mode, const float* scale, int B, int T, int C, double* sums, float* gch, void* dy, void* stream) {
    if (int e = dmx_check("dmx_act_bwd", B, T, C, mode, stats ? G : 1)) return e;
    SEHIP_REQUIRE(!scale || stats, "dmx_act_bwd: LayerScale is only built behind a GroupNorm");
    const int nq = (mode ? C >> 1 : C) >> 3;
    const dim3 g = dmx_grid(B, T, nq);
    float* part = nullptr;
    int nrows = 0;
    if (stats) {
        SEHIP_REQUIRE(gamma && beta && sums && gch, "dmx_act_bwd: GroupNorm needs gamma, beta, sums and the gradient accumulator");
        const int NV = 2 * C + (mode ? C >> 1 : C);
        static const int lds_pad = getenv("SEHIP_DMX_LDS_PAD") ? atoi(getenv("SEHIP_DMX_LDS_PAD")) : 0;       // tools/dev: unused bytes behind the partials
        const size_t lds = ((size_t)NV + 16) * sizeof(float) + (size_t)lds_pad;
        SEHIP_REQUIRE(lds <= 64 * 1024, "dmx_act_bwd: C=%d does not fit the LDS partials", C);
        // one row of per-channel sums per workgroup (per-stream pool of csrc/wgrad3.hip; none available -- e.g. a capture that would
        // have to grow it -- or SEHIP_DMX_ATOMIC_FLUSH: the workgroups flush with atomics as in rounds 2-4)
        static const bool atomic_flush = getenv("SEHIP_DMX_ATOMIC_FLUSH") != nullptr;
        nrows = (int)(g.x * g.y);
        part = atomic_flush && !sehip_deterministic() ? nullptr : sehip_wgrad_scratch((hipStream_t)stream, (size_t)nrows * NV * sizeof(float));
        SEHIP_REQUIRE(part || !sehip_deterministic(), "dmx_act_bwd: the deterministic schedule could not get its partial rows (allocation failed, "
                                                      "or inside a stream capture)");
        bool ok;
        const DetCtx dc = sehip_det_ctx((hipStream_t)stream, (size_t)nrows * 2 * G, &ok);
        if (!ok) return -2;
        if (mode)
            dmx_act_bwd_reduce_kernel<1><<<g, 256, lds, (hipStream_t)stream>>>((const bf16_raw*)dz, (const bf16_raw*)y, stats, gamma, beta, G, eps, scale, T,
                                                                                C, sums, gch, part, dc);
        else
            dmx_act_bwd_reduce_kernel<0><<<g, 256, lds, (hipStream_t)stream>>>((const bf16_raw*)dz, (const bf16_raw*)y, stats, gamma, beta, G, eps, scale, T,
                                                                                C, sums, gch, part, dc);
        SEHIP_CHECK_LAUNCH("dmx_act_bwd_reduce");
        if (int e = sehip_det_finish((hipStream_t)stream, dc, B, (int)g.x, 2 * G, sums, 2L * G)) return e;
    }
    if (mode)
        dmx_act_bwd_apply_kernel<1><<<g, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)dz, (const bf16_raw*)y, stats, sums, gamma, beta, G, eps, scale, T, C,
                                                                         (bf16_raw*)dy, part, nrows, gch, sehip_deterministic());
    else
        dmx_act_bwd_apply_kernel<0><<<g, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)dz, (const bf16_raw*)y, stats, sums, gamma, beta, G, eps, scale, T, C,
                                                                         (bf16_raw*)dy, part, nrows, gch, sehip_deterministic());
    SEHIP_CHECK_LAUNCH("dmx_act_bwd_apply");
    return 0;
}

extern "C" int sehip_dmx_add(const void* a, const void* b, long n, void* out, void* stream) {
    SEHIP_REQUIRE(n > 0 && (n & 7) == 0, "dmx_add: n=%ld must be a positive multiple of 8", n);
    long g = (n / 8 + 255) / 256;
    if (g > 4096) g = 4096;
    dmx_add_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)a, (const bf16_raw*)b, n / 8, (bf16_raw*)out);
    SEHIP_CHECK_LAUNCH("dmx_add");
    return 0;
}

extern "C" int sehip_dmx_frames(int mode, const void* a, const void* b, int B, int T, int C, int nf, int W, int S, void* out, void* stream) {
    SEHIP_REQUIRE(mode >= 0 && mode <= 3 && B > 0 && T > 0 && C >= 8 && (C & 7) == 0 && nf >= 1 && W >= 1 && S >= 1,
                  "dmx_frames: bad arguments (mode=%d B=%d T=%d C=%d nf=%d W=%d S=%d)", mode, B, T, C, nf, W, S);
    SEHIP_REQUIRE(nf == 1 ? W == T : (W == 2 * S && (S & 1) == 0 && (long)S * (nf - 1) < T && T <= (long)S * nf),
                  "dmx_frames: %d chunks of %d frames at hop %d do not tile %d frames", nf, W, S, T);
    SEHIP_REQUIRE(mode == 0 || mode == 2 || b, "dmx_frames: modes 1 and 3 add a second tensor");
    const long total = ((mode == 0 || mode == 2) ? (long)B * nf * W : (long)B * T) * (C >> 3);
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    dmx_frames_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(mode, (const bf16_raw*)a, (const bf16_raw*)b, B, T, C, nf, W, S, (bf16_raw*)out);
    SEHIP_CHECK_LAUNCH("dmx_frames");
    return 0;
}

extern "C" int sehip_dmx_f32_to_bf16(const float* a, long n, void* out, void* stream) {
    SEHIP_REQUIRE(n > 0, "dmx_f32_to_bf16: empty");
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    dmx_f32_to_bf16_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(a, n, (bf16_raw*)out);
    SEHIP_CHECK_LAUNCH("dmx_f32_to_bf16");
    return 0;
}

// the whole layer: ONE persistent launch where its workgroups can all be resident, otherwise (or with sync == NULL) T step launches
template <int KS>
static void lstm_seq_fwd_launch(dim3 g, hipStream_t st, float* pre, const void* whh, void* hs, float* cs, int Bn, int T, int btiles, unsigned* sync) {
    dmx_lstm_seq_fwd_kernel<KS><<<g, 256, 0, st>>>(pre, (const bf16_raw*)whh, (bf16_raw*)hs, cs, Bn, T, btiles, sync);
}
template <int KS>
static void lstm_seq_bwd_launch(dim3 g, hipStream_t st, const float* gates, const void* whhT, const float* cs, const void* dhs, void* dG, int Bn, int T,
                                int btiles, unsigned* sync) {
    dmx_lstm_seq_bwd_kernel<KS><<<g, 256, 0, st>>>(gates, (const bf16_raw*)whhT, cs, (const bf16_raw*)dhs, (bf16_raw*)dG, Bn, T, btiles, sync);
}
// Co-residency bound of the spinning kernels: every workgroup of the grid must be resident at once (the members of a group wait
// for each other), and they share the GPU with the weight-gradient, packing and communication streams.  The bound is what the
// runtime reports for the kernel (workgroups per CU x CUs), halved: the other streams' workgroups may hold a CU's LDS / registers
// when a member is dispatched, and a member that is dispatched late only makes the others spin, while one that cannot be
// dispatched until they finish is a time-out (sehip_dmx_lstm_* then report it through the sticky word, plan_demucs falls back).
template <typename K>
static int lstm_seq_capacity(K kernel) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess) return 0;
    return per_cu * cus / 2;
}
static int lstm_seq_capacity_for(int ks, bool bwd) {
    static int cap[2][5] = {{-1, -1, -1, -1, -1}, {-1, -1, -1, -1, -1}};
    const int i = ks == 1 ? 0 : ks == 2 ? 1 : ks == 4 ? 2 : ks == 8 ? 3 : 4;
    int& c = cap[bwd ? 1 : 0][i];
    if (c < 0) {
        switch (ks * 2 + (bwd ? 1 : 0)) {
            case 2: c = lstm_seq_capacity(dmx_lstm_seq_fwd_kernel<1>); break;
            case 3: c = lstm_seq_capacity(dmx_lstm_seq_bwd_kernel<1>); break;
            case 4: c = lstm_seq_capacity(dmx_lstm_seq_fwd_kernel<2>); break;
            case 5: c = lstm_seq_capacity(dmx_lstm_seq_bwd_kernel<2>); break;
            case 8: c = lstm_seq_capacity(dmx_lstm_seq_fwd_kernel<4>); break;
            case 9: c = lstm_seq_capacity(dmx_lstm_seq_bwd_kernel<4>); break;
            case 16: c = lstm_seq_capacity(dmx_lstm_seq_fwd_kernel<8>); break;
            case 17: c = lstm_seq_capacity(dmx_lstm_seq_bwd_kernel<8>); break;
            case 32: c = lstm_seq_capacity(dmx_lstm_seq_fwd_kernel<16>); break;
            default: c = lstm_seq_capacity(dmx_lstm_seq_bwd_kernel<16>); break;
        }
    }
    return c;
}
static bool lstm_seq_ok(int Bn, int T, int H, const unsigned* sync, bool bwd) {
    static const bool off = getenv("SEHIP_DMX_LSTM_STEPS") != nullptr;
    const int ks = H / 32, btiles = (Bn + 15) / 16;
    if (!sync || off || !(ks == 1 || ks == 2 || ks == 4 || ks == 8 || ks == 16)) return false;
    const int grid = 8 * (H / 16) * ((2 * btiles + 7) / 8);          // incl. the inactive workgroups that pad a group to one XCD
    return grid <= lstm_seq_capacity_for(ks, bwd) && DMX_CNT0 + 2 * btiles * DMX_REPL * DMX_RSTRIDE <= DMX_SYNC_WORDS &&
           (long)Bn * T * 8 * H * 2 < (1L << 31);      // 32-bit byte offsets into the handed-off tensors
}
extern "C" int sehip_dmx_lstm_sync_bytes(void) { return DMX_SYNC_WORDS * 4; }

extern "C" int sehip_dmx_lstm_fwd(float* pre, const void* whh, int Bn, int T, int H, void* hs, float* cs, unsigned* sync, void* stream) {
    SEHIP_REQUIRE(Bn > 0 && T > 0 && H >= 32 && (H & 31) == 0, "dmx_lstm_fwd: hidden size H=%d must be a multiple of 32 (Bn=%d T=%d)", H, Bn, T);
    hipStream_t st = (hipStream_t)stream;
    const int btiles = (Bn + 15) / 16;
    if (lstm_seq_ok(Bn, T, H, sync, false)) {
        SEHIP_REQUIRE(hipMemsetAsync(sync + DMX_CNT0, 0, (size_t)2 * btiles * DMX_REPL * DMX_RSTRIDE * 4, st) == hipSuccess, "dmx_lstm_fwd: clearing the sync block failed");
        const dim3 g(8 * (H / 16) * ((2 * btiles + 7) / 8));
        switch (H / 32) {
            case 1: lstm_seq_fwd_launch<1>(g, st, pre, whh, hs, cs, Bn, T, btiles, sync); break;
            case 2: lstm_seq_fwd_launch<2>(g, st, pre, whh, hs, cs, Bn, T, btiles, sync); break;
            case 4: lstm_seq_fwd_launch<4>(g, st, pre, whh, hs, cs, Bn, T, btiles, sync); break;
            case 8: lstm_seq_fwd_launch<8>(g, st, pre, whh, hs, cs, Bn, T, btiles, sync); break;
            default: lstm_seq_fwd_launch<16>(g, st, pre, whh, hs, cs, Bn, T, btiles, sync); break;
        }
        SEHIP_CHECK_LAUNCH("dmx_lstm_seq_fwd");
        return 0;
    }
    const dim3 g(H / 16, 2, btiles);
    for (int s = 0; s < T; ++s)
        dmx_lstm_step_fwd_kernel<<<g, 256, 0, st>>>(pre, (const bf16_raw*)whh, (bf16_raw*)hs, cs, Bn, T, H, s);
    SEHIP_CHECK_LAUNCH("dmx_lstm_fwd");
    return 0;
}

extern "C" int sehip_dmx_lstm_bwd(const float* gates, const void* whhT, const float* cs, const void* dhs, int Bn, int T, int H, void* dG, float* dc,
                                  unsigned* sync, void* stream) {
    SEHIP_REQUIRE(Bn > 0 && T > 0 && H >= 32 && (H & 31) == 0, "dmx_lstm_bwd: hidden size H=%d must be a multiple of 32", H);
    hipStream_t st = (hipStream_t)stream;
    const int btiles = (Bn + 15) / 16;
    if (lstm_seq_ok(Bn, T, H, sync, true)) {
        SEHIP_REQUIRE(hipMemsetAsync(sync + DMX_CNT0, 0, (size_t)2 * btiles * DMX_REPL * DMX_RSTRIDE * 4, st) == hipSuccess, "dmx_lstm_bwd: clearing the sync block failed");
        const dim3 g(8 * (H / 16) * ((2 * btiles + 7) / 8));
        switch (H / 32) {
            case 1: lstm_seq_bwd_launch<1>(g, st, gates, whhT, cs, dhs, dG, Bn, T, btiles, sync); break;
            case 2: lstm_seq_bwd_launch<2>(g, st, gates, whhT, cs, dhs, dG, Bn, T, btiles, sync); break;
            case 4: lstm_seq_bwd_launch<4>(g, st, gates, whhT, cs, dhs, dG, Bn, T, btiles, sync); break;
            case 8: lstm_seq_bwd_launch<8>(g, st, gates, whhT, cs, dhs, dG, Bn, T, btiles, sync); break;
            default: lstm_seq_bwd_launch<16>(g, st, gates, whhT, cs, dhs, dG, Bn, T, btiles, sync); break;
        }
        SEHIP_CHECK_LAUNCH("dmx_lstm_seq_bwd");
        return 0;
    }
    SEHIP_REQUIRE(dc, "dmx_lstm_bwd: the per-step path needs the cell-gradient scratch");
    const dim3 g(H / 16, 2, btiles);
    for (int s = 0; s < T; ++s)
        dmx_lstm_step_bwd_kernel<<<g, 256, 0, st>>>(gates, (const bf16_raw*)whhT, cs, (const bf16_raw*)dhs, (bf16_raw*)dG, dc, Bn, T, H, s);
    SEHIP_CHECK_LAUNCH("dmx_lstm_bwd");
    return 0;
}

static int attn_check(const char* who, int B, int T, int hid, int heads, int nd, int NQ, size_t lds) {
    SEHIP_REQUIRE(B > 0 && T > 0 && heads > 0 && hid % heads == 0 && ((hid / heads) & 7) == 0,
                  "%s: %d channels over %d heads: a head must hold a multiple of 8 channels", who, hid, heads);
    SEHIP_REQUIRE(nd >= 0 && nd <= 8 && NQ >= 3 * hid + heads * nd && (NQ & 7) == 0, "%s: bad row length NQ=%d", who, NQ);
    SEHIP_REQUIRE(lds <= 150 * 1024, "%s: T=%d frames do not fit the LDS score tile", who, T);
    return 0;
}

extern "C" int sehip_dmx_attn_fwd(const void* qkv, int B, int T, int hid, int heads, int nd, int NQ, void* out, void* stream) {
    const size_t base = ((size_t)T * QT + QT + 8 * QT) * sizeof(float);
    if (int e = attn_check("dmx_attn_fwd", B, T, hid, heads, nd, NQ, base)) return e;
    const size_t tiles = ((size_t)2 * T + QT) * (hid / heads + 8) * sizeof(bf16_raw);
    const int stage = base + tiles <= 150 * 1024;
    const dim3 grid((T + QT - 1) / QT, heads, B);
    const size_t lds = stage ? base + tiles : base;
#define DMX_ATTN_FWD(NCH)                                                                                                              \
    do {                                                                                                                               \
        static bool attr = false;                                                                                                      \
        if (!attr) { (void)hipFuncSetAttribute((const void*)dmx_attn_fwd_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; } \
        dmx_attn_fwd_kernel<NCH><<<grid, 256, lds, (hipStream_t)stream>>>((const bf16_raw*)qkv, T, hid, heads, nd, NQ, stage, (bf16_raw*)out);     \
    } while (0)
    switch (hid / heads / 8) {
        case 1: DMX_ATTN_FWD(1); break;
        case 2: DMX_ATTN_FWD(2); break;
        case 4: DMX_ATTN_FWD(4); break;
        case 8: DMX_ATTN_FWD(8); break;
        case 16: DMX_ATTN_FWD(16); break;
        default: DMX_ATTN_FWD(0); break;
    }
#undef DMX_ATTN_FWD
    SEHIP_CHECK_LAUNCH("dmx_attn_fwd");
    return 0;
}

extern "C" long sehip_dmx_attn_bwd_scratch_floats(int B, int T, int hid) { return (long)((T + QT - 1) / QT) * B * T * 2 * hid; }

extern "C" int sehip_dmx_attn_bwd(const void* qkv, const void* dres, int B, int T, int hid, int heads, int nd, int NQ, float* slabs, void* dqkv_bf16,
                                  void* stream) {
    const size_t base = ((size_t)2 * T * QT + QT + 8 * QT) * sizeof(float);
    if (int e = attn_check("dmx_attn_bwd", B, T, hid, heads, nd, NQ, base)) return e;
    SEHIP_REQUIRE(slabs && dqkv_bf16, "dmx_attn_bwd: null output");
    const size_t tiles = ((size_t)2 * T + 2 * QT) * (hid / heads + 8) * sizeof(bf16_raw);
    const int stage = base + tiles <= 150 * 1024;
    const dim3 grid((T + QT - 1) / QT, heads, B);
    const size_t lds = stage ? base + tiles : base;
#define DMX_ATTN_BWD(NCH)                                                                                                              \
    do {                                                                                                                               \
        static bool attr = false;                                                                                                      \
        if (!attr) { (void)hipFuncSetAttribute((const void*)dmx_attn_bwd_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr = true; } \
        dmx_attn_bwd_kernel<NCH><<<grid, 256, lds, (hipStream_t)stream>>>((const bf16_raw*)qkv, (const bf16_raw*)dres, T, hid, heads, nd, NQ, stage, slabs, \
                                                                          (bf16_raw*)dqkv_bf16);                                       \
    } while (0)
    switch (hid / heads / 8) {
        case 1: DMX_ATTN_BWD(1); break;
        case 2: DMX_ATTN_BWD(2); break;
        case 4: DMX_ATTN_BWD(4); break;
        case 8: DMX_ATTN_BWD(8); break;
        case 16: DMX_ATTN_BWD(16); break;
        default: DMX_ATTN_BWD(0); break;
    }
#undef DMX_ATTN_BWD
    SEHIP_CHECK_LAUNCH("dmx_attn_bwd");
    const long items = (long)B * T * ((2 * hid) >> 3);
    long g = (items + 255) / 256;
    if (g > 2048) g = 2048;
    dmx_attn_bwd_sum_kernel<<<(unsigned)g, 256, 0, (hipStream_t)stream>>>(slabs, (T + QT - 1) / QT, (long)B * T, hid, NQ, (bf16_raw*)dqkv_bf16);
    SEHIP_CHECK_LAUNCH("dmx_attn_bwd_sum");
    return 0;
}
