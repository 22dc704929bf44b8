// Front and back end of the complex DCUnet (src/model/dcunet.py:102-162) around its convolution stack:
//   dcunet_pack_input : the STFT-domain input [R][F][T][2] fp32 (what stft_custom returns, time innermost) ->
//                       channels-last bf16 [R][T][F][2] = x.transpose(2, 3) of :106, the first encoder's source
//   dcunet_mask_fwd   : linear = 1x1 ComplexConv2d(C -> 1) (:93-95, :323-338) + tanh (:131) + transpose back (:132)
//                       + the mask of :136-159 applied to the input spectrum -> enhanced spectrum [R][F][T][2] fp32
//   dcunet_mask_bwd   : d enhanced -> d(last decoder output) [R][T][F][2 Cs] bf16 + gradients of the 1x1 conv
// All three are HBM streams: the last decoder's output is 62 complex channels at the FULL 257 x 257 resolution
// (1.08 GB in bf16 at batch 64); the 1x1 convolution reads it exactly once (a row is 256 contiguous bytes, 16 lanes
// x 16 bytes), the backward pass reads it once more for the weight gradient while it writes its gradient.
// Tiles of 32 frames x 32 bins go through LDS so that both the [T][F] side (activations) and the [F][T] side
// (spectra) are accessed in >= 128-byte runs.
#include "common.h"
float* sehip_wgrad_scratch(hipStream_t st, size_t bytes);   // csrc/wgrad3.hip: per-stream pool of partial arrays
#include "mask.h"
#include "rbn.h"

#define DT 32   // tile edge (frames and bins)
#define DP 33   // LDS pitch of a tile row (float2 elements)

__global__ __launch_bounds__(256) void dcunet_pack_input_kernel(const float2* __restrict__ spec, int F, int T,
                                                                unsigned* __restrict__ out) {
    __shared__ float2 tile[DT][DP];
    const int r = blockIdx.z, t0 = blockIdx.x * DT, f0 = blockIdx.y * DT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float2* s = spec + (size_t)r * F * T;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = f0 + ty + 8 * k, t = t0 + tx;
        if (f < F && t < T) tile[ty + 8 * k][tx] = s[(size_t)f * T + t];
    }
    __syncthreads();
    unsigned* o = out + (size_t)r * T * F;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, f = f0 + tx;
        if (f < F && t < T) {
            const float2 v = tile[tx][ty + 8 * k];
            o[(size_t)t * F + f] = pack_bf2(v.x, v.y);
        }
    }
}

// per-lane coefficients of the 1x1 complex conv for this lane's 8-channel piece q of a row (nq = C/8 pieces; the first
// nq/2 are real-part channels, the rest imaginary-part channels):
//   lin_re = sum wre zr - wim zi ; lin_im = sum wre zi + wim zr    ->  piece of zr: (wa, wb) = (wre, wim); of zi: (-wim, wre)
struct LinCoef { float wa[8], wb[8]; };
__device__ __forceinline__ LinCoef lin_coef(const float* __restrict__ w_re, const float* __restrict__ w_im, int q, int nq, int Cs,
                                            int Cr) {
    LinCoef c;
    const bool imag = q >= (nq >> 1);
    const int c0 = (imag ? q - (nq >> 1) : q) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool ok = c0 + j < Cr;
        const float wr = ok ? w_re[c0 + j] : 0.f, wi = ok ? w_im[c0 + j] : 0.f;
        c.wa[j] = imag ? -wi : wr;
        c.wb[j] = imag ? wr : wi;
    }
    return c;
}

__device__ __forceinline__ void unpack8f(uint4 u, float (&v)[8]) {
    v[0] = bf2f((bf16_raw)(u.x & 0xffff)); v[1] = bf2f((bf16_raw)(u.x >> 16));
    v[2] = bf2f((bf16_raw)(u.y & 0xffff)); v[3] = bf2f((bf16_raw)(u.y >> 16));
    v[4] = bf2f((bf16_raw)(u.z & 0xffff)); v[5] = bf2f((bf16_raw)(u.z >> 16));
    v[6] = bf2f((bf16_raw)(u.w & 0xffff)); v[7] = bf2f((bf16_raw)(u.w >> 16));
}

__global__ __launch_bounds__(256) void dcunet_mask_fwd_kernel(const bf16_raw* __restrict__ z, const float* __restrict__ w_re,
                                                              const float* __restrict__ w_im, const float* __restrict__ b_re,
                                                              const float* __restrict__ b_im, const float2* __restrict__ spec,
                                                              int F, int T, int Cs, int Cr, int mode, float2* __restrict__ mask_ws,
                                                              float2* __restrict__ out, const float4* __restrict__ coef) {
    // coef != NULL (fused tail): z is the last decoder's PRE-BatchNorm output and coef its per-channel (scale, shift, mean, rstd) of
    // csrc/rbn.hip: BatchNorm + LeakyReLU are applied to the loaded piece, the activation tensor is never written
    __shared__ float2 tile[DT][DP];   // tanh(linear) per position, [frame][bin]
    const int r = blockIdx.z, t0 = blockIdx.x * DT, f0 = blockIdx.y * DT;
    const int C = 2 * Cs, nq = C >> 3, ppi = 64 / nq;   // positions per wave instruction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % nq, pl = lane / nq;
    const LinCoef lc = lin_coef(w_re, w_im, q, nq, Cs, Cr);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 k = coef ? coef[q * 8 + j] : make_float4(1.f, 0.f, 0.f, 0.f);
        sc[j] = k.x; sh[j] = k.y;
    }
    const bool bn = coef != nullptr;
    const float bre = b_re[0] - b_im[0], bim = b_re[0] + b_im[0];   // each of the four real convs carries its own bias
    const bf16_raw* zb = z + (size_t)r * T * F * C;
    const int fw = min(DT, F - f0);            // valid bins of this tile
    const int npos = min(DT, T - t0) * fw;
    // four trips per pass: their loads are issued before the first use (one 16-byte load per trip in flight was 3.2 TB/s on the
    // 1.08-GB tensor of the C2 shape; BatchNorm's statistics pass reads the same tensor at 5.6)
    for (int pb = wave * ppi; pb < npos; pb += 16 * ppi) {
        uint4 u[4];
        int tl[4], fl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = pb + 4 * ppi * k + pl;
            tl[k] = 0; fl[k] = 0; u[k] = make_uint4(0u, 0u, 0u, 0u);
            if (p < npos) {
                tl[k] = p / fw; fl[k] = p - tl[k] * fw;
                u[k] = *reinterpret_cast<const uint4*>(zb + ((size_t)(t0 + tl[k]) * F + f0 + fl[k]) * C + q * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = pb + 4 * ppi * k + pl;
            float sre = 0.f, sim = 0.f;
            if (p < npos) {
                float v[8];
                unpack8f(u[k], v);
                if (bn) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float o = sc[j] * v[j] + sh[j]; v[j] = o > 0.f ? o : RBN_SLOPE * o; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { sre += lc.wa[j] * v[j]; sim += lc.wb[j] * v[j]; }
            }
            for (int o = 1; o < nq; o <<= 1) { sre += __shfl_xor(sre, o, 64); sim += __shfl_xor(sim, o, 64); }
            if (p < npos && q == 0) tile[tl[k]][fl[k]] = make_float2(tanhf(sre + bre), tanhf(sim + bim));
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float2* mw = mask_ws + (size_t)r * T * F;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // the mask, [frame][bin] order, kept for the backward pass
        const int t = t0 + ty + 8 * k, f = f0 + tx;
        if (t < T && f < F) mw[(size_t)t * F + f] = tile[ty + 8 * k][tx];
    }
    const float2* s = spec + (size_t)r * F * T;
    float2* ob = out + (size_t)r * F * T;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // masking, [bin][frame] order
        const int f = f0 + ty + 8 * k, t = t0 + tx;
        if (t < T && f < F) {
            const float2 x = s[(size_t)f * T + t], m = tile[tx][ty + 8 * k];
            float er, ei;
            apply_mask(mode, x.x, x.y, m.x, m.y, er, ei);
            ob[(size_t)f * T + t] = make_float2(er, ei);
        }
    }
}

// gacc: fp32 [2*Cs + 2]: d w_re [Cs], d w_im [Cs], d b_re, d b_im (caller zeroes; accumulated with atomics)
__global__ __launch_bounds__(256) void dcunet_mask_bwd_kernel(const float2* __restrict__ dout, const float2* __restrict__ spec,
                                                              const float2* __restrict__ mask_ws, const bf16_raw* __restrict__ z,
                                                              const float* __restrict__ w_re, const float* __restrict__ w_im, int F,
                                                              int T, int Cs, int Cr, int mode, bf16_raw* __restrict__ dz,
                                                              float* __restrict__ gacc, float* __restrict__ part) {
    // part != NULL (deterministic schedule): this workgroup's 2 Cs + 2 sums go to row (z, y, x) of `part` by plain stores and
    // dcunet_rows_reduce_kernel adds the rows in order, instead of one fp32 atomic per weight and workgroup
    __shared__ float2 tile[DT][DP];      // in: the mask; out: d linear, [frame][bin]
    __shared__ float red[4][32][16];     // per wave: [piece q < nq <= 32][8 a-sums | 8 b-sums]
    const int r = blockIdx.z, t0 = blockIdx.x * DT, f0 = blockIdx.y * DT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float2* mw = mask_ws + (size_t)r * T * F;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, f = f0 + tx;
        if (t < T && f < F) tile[ty + 8 * k][tx] = mw[(size_t)t * F + f];
    }
    __syncthreads();
    const float2* s = spec + (size_t)r * F * T;
    const float2* go = dout + (size_t)r * F * T;
    float2 gl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = f0 + ty + 8 * k, t = t0 + tx;
        gl[k] = make_float2(0.f, 0.f);
        if (t < T && f < F) {
            const float2 x = s[(size_t)f * T + t], g = go[(size_t)f * T + t], m = tile[tx][ty + 8 * k];
            float gmr, gmi;
            mask_grad(mode, x.x, x.y, m.x, m.y, g.x, g.y, gmr, gmi);
            gl[k] = make_float2(gmr * (1.f - m.x * m.x), gmi * (1.f - m.y * m.y));   // through the tanh
        }
    }
    float sbr = 0.f, sbi = 0.f;   // bias gradients: d b_re = sum (g_re + g_im), d b_im = sum (-g_re + g_im)
#pragma unroll
    for (int k = 0; k < 4; ++k) { sbr += gl[k].x + gl[k].y; sbi += gl[k].y - gl[k].x; }
    __syncthreads();              // every thread has read its mask values: the tile is rewritten with d linear
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[tx][ty + 8 * k] = gl[k];
    __syncthreads();

    const int C = 2 * Cs, nq = C >> 3, ppi = 64 / nq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane % nq, pl = lane / nq;
    const LinCoef lc = lin_coef(w_re, w_im, q, nq, Cs, Cr);
    const bf16_raw* zb = z + (size_t)r * T * F * C;
    bf16_raw* dzb = dz + (size_t)r * T * F * C;
    const int fw = min(DT, F - f0);
    const int npos = min(DT, T - t0) * fw;
    float aa[8], ab[8];           // sum g_re z[c], sum g_im z[c] over this lane's positions
#pragma unroll
    for (int j = 0; j < 8; ++j) { aa[j] = 0.f; ab[j] = 0.f; }
    for (int p0 = wave * ppi; p0 < npos; p0 += 4 * ppi) {
        const int p = p0 + pl;
        if (p >= npos) continue;
        const int tl = p / fw, fl = p - tl * fw;
        const float2 g = tile[tl][fl];
        const size_t off = ((size_t)(t0 + tl) * F + f0 + fl) * C + q * 8;
        float v[8], o[8];
        unpack8f(*reinterpret_cast<const uint4*>(zb + off), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = g.x * lc.wa[j] + g.y * lc.wb[j];
            aa[j] += g.x * v[j]; ab[j] += g.y * v[j];
        }
        *reinterpret_cast<uint4*>(dzb + off) = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    }
    // fold the lanes that own the same piece, then the four waves, then one atomic per weight and workgroup
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        for (int o = nq; o < 64; o <<= 1) { aa[j] += __shfl_xor(aa[j], o, 64); ab[j] += __shfl_xor(ab[j], o, 64); }
    }
    if (lane < nq) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[wave][lane][j] = aa[j]; red[wave][lane][8 + j] = ab[j]; }
    }
    sbr = wave_sum(sbr); sbi = wave_sum(sbi);
    __shared__ float redb[4][2];
    if (lane == 0) { redb[wave][0] = sbr; redb[wave][1] = sbi; }
    __syncthreads();
    if (threadIdx.x < nq * 8) {
        const int qq = threadIdx.x >> 3, j = threadIdx.x & 7;
        const float a = red[0][qq][j] + red[1][qq][j] + red[2][qq][j] + red[3][qq][j];          // sum g_re z
        const float b = red[0][qq][8 + j] + red[1][qq][8 + j] + red[2][qq][8 + j] + red[3][qq][8 + j];  // sum g_im z
        const bool imag = qq >= (nq >> 1);
        const int c = (imag ? qq - (nq >> 1) : qq) * 8 + j;
        if (c < Cr) {
            // real-part piece: d wre += g_re zr, d wim += g_im zr ; imaginary-part piece: d wre += g_im zi, d wim -= g_re zi
            if (part) {          // [row][2 halves (real-part piece, imaginary-part piece)][2 Cs + 2]: the reduction adds the halves too
                float* pr = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)(2 * (2 * Cs + 2)) +
                            (imag ? 2 * Cs + 2 : 0);
                pr[c] = imag ? b : a;
                pr[Cs + c] = imag ? -a : b;
            } else {
                atomicAdd(&gacc[c], imag ? b : a);
                atomicAdd(&gacc[Cs + c], imag ? -a : b);
            }
        }
    }
    if (threadIdx.x == 0) {
        const float s0 = redb[0][0] + redb[1][0] + redb[2][0] + redb[3][0], s1 = redb[0][1] + redb[1][1] + redb[2][1] + redb[3][1];
        if (part) {
            float* pr = part + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (size_t)(2 * (2 * Cs + 2));
            pr[2 * Cs] = s0; pr[2 * Cs + 1] = s1;
            pr[2 * Cs + 2 + 2 * Cs] = 0.f; pr[2 * Cs + 2 + 2 * Cs + 1] = 0.f;
        } else {
            atomicAdd(&gacc[2 * Cs], s0);
            atomicAdd(&gacc[2 * Cs + 1], s1);
        }
    }
}

// deterministic schedule: gacc[i] += sum over the rows, in row order (both halves of a row), i < 2 Cs + 2; entries with c >= Cr of the
// weight halves were never written and are skipped
__global__ __launch_bounds__(256) void dcunet_rows_reduce_kernel(const float* __restrict__ part, int nrows, int Cs, int Cr,
                                                                 float* __restrict__ gacc) {
    const int i = blockIdx.x * 256 + threadIdx.x, w = 2 * Cs + 2;
    if (i >= w) return;
    if (i < 2 * Cs && (i % Cs) >= Cr) return;
    float acc = 0.f;
    for (int r = 0; r < nrows; ++r) acc += part[(size_t)r * 2 * w + i] + part[(size_t)r * 2 * w + w + i];
    gacc[i] += acc;
}

// ---- the fused tail of the backward pass --------------------------------------------------------------------------------------
// d(last decoder output) = the 1x1 convolution's transpose applied to d linear: TWO values per position spread over 2 Cs
// channels.  The unfused path writes that 1.08 GB tensor (B = 64), BatchNorm's backward reads it twice with the pre-activation
// and writes the gradient.  Here d linear stays the only thing stored per position ([R][T][F] float2, over the mask workspace),
// and the two BatchNorm passes rebuild the row from it:
//   dlin     tile kernel over the spectra: d enhanced -> d linear (through mask and tanh), [frame][bin] order; bias partials
//   reduce   rows of y: per channel sum g, sum g xh (BatchNorm), sum g_re z, sum g_im z (the 1x1 conv's weights)
//   finalize per channel: BatchNorm weight / bias gradients + its backward coefficients, the 1x1 conv's weight and bias gradients
//   apply    rows of y: dy = w rstd (g - mean g - xh mean(g xh))
// y is read twice, dy written once: 3.3 GB instead of 7.6 GB; nothing is accumulated with atomics in an order that varies (the
// two addends per 1x1 weight commute), so the deterministic schedule runs the same kernels.
__global__ __launch_bounds__(256) void dcunet_dlin_kernel(const float2* __restrict__ dout, const float2* __restrict__ spec,
                                                          float2* __restrict__ mask_ws, int F, int T, int mode,
                                                          float* __restrict__ bpart) {
    __shared__ float2 tile[DT][DP];
    __shared__ float redb[4][2];
    const int r = blockIdx.z, t0 = blockIdx.x * DT, f0 = blockIdx.y * DT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float2* mw = mask_ws + (size_t)r * T * F;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, f = f0 + tx;
        if (t < T && f < F) tile[ty + 8 * k][tx] = mw[(size_t)t * F + f];
    }
    __syncthreads();
    const float2* s = spec + (size_t)r * F * T;
    const float2* go = dout + (size_t)r * F * T;
    float2 gl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = f0 + ty + 8 * k, t = t0 + tx;
        gl[k] = make_float2(0.f, 0.f);
        if (t < T && f < F) {
            const float2 x = s[(size_t)f * T + t], g = go[(size_t)f * T + t], m = tile[tx][ty + 8 * k];
            float gmr, gmi;
            mask_grad(mode, x.x, x.y, m.x, m.y, g.x, g.y, gmr, gmi);
            gl[k] = make_float2(gmr * (1.f - m.x * m.x), gmi * (1.f - m.y * m.y));   // through the tanh
        }
    }
    float sbr = 0.f, sbi = 0.f;   // bias gradients: d b_re = sum (g_re + g_im), d b_im = sum (-g_re + g_im)
#pragma unroll
    for (int k = 0; k < 4; ++k) { sbr += gl[k].x + gl[k].y; sbi += gl[k].y - gl[k].x; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[tx][ty + 8 * k] = gl[k];
    sbr = wave_sum(sbr); sbi = wave_sum(sbi);
    if ((threadIdx.x & 63) == 0) { redb[threadIdx.x >> 6][0] = sbr; redb[threadIdx.x >> 6][1] = sbi; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = t0 + ty + 8 * k, f = f0 + tx;
        if (t < T && f < F) mw[(size_t)t * F + f] = tile[ty + 8 * k][tx];
    }
    if (threadIdx.x == 0) {
        float* pr = bpart + 2 * (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        pr[0] = redb[0][0] + redb[1][0] + redb[2][0] + redb[3][0];
        pr[1] = redb[0][1] + redb[1][1] + redb[2][1] + redb[3][1];
    }
}

__global__ __launch_bounds__(256) void dcunet_tail_reduce_kernel(const float2* __restrict__ dlin, const bf16_raw* __restrict__ y,
                                                                 const float4* __restrict__ coef, const float* __restrict__ w_re,
                                                                 const float* __restrict__ w_im, long rows, int Cs, int Cr,
                                                                 float* __restrict__ part) {
    __shared__ float lds[4 * 4 * 8 * 32];        // [4 waves][NS * 8][nq <= 32]
    const int C = 2 * Cs, nq = C >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const LinCoef lc = lin_coef(w_re, w_im, q, nq, Cs, Cr);
    float4 k[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) k[j] = coef[q * 8 + j];
    float s[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[a][j] = 0.f;
    const long stride = (long)gridDim.x * rpb;
    for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 4 * stride) {
        uint4 uy[4];
        float2 g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long r = r0 + t * stride;
            uy[t] = make_uint4(0u, 0u, 0u, 0u); g[t] = make_float2(0.f, 0.f);
            if (r < rows) { uy[t] = *reinterpret_cast<const uint4*>(y + r * C + q * 8); g[t] = dlin[r]; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const RChunk8 a = r_unpack8(uy[t]);       // rows beyond the end: g == 0 contributes nothing
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float o = k[j].x * a.v[j] + k[j].y;
                const bool pos = o > 0.f;
                const float z = pos ? o : RBN_SLOPE * o;
                const float dzv = g[t].x * lc.wa[j] + g[t].y * lc.wb[j];
                const float gg = pos ? dzv : RBN_SLOPE * dzv;
                s[0][j] += gg;
                s[1][j] += gg * (a.v[j] - k[j].z) * k[j].w;
                s[2][j] += g[t].x * z;
                s[3][j] += g[t].y * z;
            }
        }
    }
    rbn_block_partials<4>(s, nq, C, part, lds);
}

__global__ void dcunet_tail_finalize_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ bpart, int nbp,
                                            const float4* __restrict__ coef, long rows, int Cs, int Cr, float* __restrict__ gw_re,
                                            float* __restrict__ gb_re, float* __restrict__ gw_im, float* __restrict__ gb_im,
                                            float4* __restrict__ bcoef, float* __restrict__ gacc) {
    const int c = blockIdx.x, C = 2 * Cs;
    const int half = c / Cs, i = c - half * Cs;
    if (c == 0) {        // the 1x1 conv's two bias gradients: the tile kernel's partials, in order
        double b0 = 0.0, b1 = 0.0;
        for (int r0 = threadIdx.x; r0 < nbp; r0 += 64 * 8) {       // eight loads in flight per lane (81 dependent trips took 30 us)
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 64 * u;
                v[u] = r < nbp ? *reinterpret_cast<const float2*>(bpart + 2 * r) : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { b0 += (double)v[u].x; b1 += (double)v[u].y; }
        }
        b0 = wave_sum_d(b0); b1 = wave_sum_d(b1);
        if (threadIdx.x == 0) { gacc[2 * Cs] += (float)b0; gacc[2 * Cs + 1] += (float)b1; }
    }
    if (i >= Cr) {
        if (threadIdx.x == 0) bcoef[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 k = coef[c];
    double a[4];
    rbn_wave_reduce<4>(part, nblk, C, c, a);
    if (threadIdx.x != 0) return;
    (half ? gb_im : gb_re)[i] = (float)a[0];
    (half ? gw_im : gw_re)[i] = (float)a[1];
    const double n = (double)rows;
    bcoef[c] = make_float4(k.x, (float)(a[0] / n), (float)(a[1] / n), 0.f);
    // real-part channel: d wre += sum g_re zr, d wim += sum g_im zr; imaginary-part channel: d wre += sum g_im zi, d wim -= sum g_re zi
    // (two addends per weight, onto the caller's zero: the order does not matter)
    atomicAdd(&gacc[i], half ? (float)a[3] : (float)a[2]);
    atomicAdd(&gacc[Cs + i], half ? -(float)a[2] : (float)a[3]);
}

__global__ __launch_bounds__(256) void dcunet_tail_apply_kernel(const float2* __restrict__ dlin, const bf16_raw* __restrict__ y,
                                                                const float4* __restrict__ coef, const float4* __restrict__ bcoef,
                                                                const float* __restrict__ w_re, const float* __restrict__ w_im,
                                                                long rows, int Cs, int Cr, bf16_raw* __restrict__ dy) {
    const int C = 2 * Cs, nq = C >> 3;
    const int q = threadIdx.x % nq, rl = threadIdx.x / nq, rpb = 256 / nq;
    const LinCoef lc = lin_coef(w_re, w_im, q, nq, Cs, Cr);
    float4 k[8];
    float ka[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        k[j] = coef[q * 8 + j];
        const float4 kb = bcoef[q * 8 + j];
        ka[j] = kb.x; k1[j] = kb.y; k2[j] = kb.z;
    }
    const long stride = (long)gridDim.x * rpb;
    for (long r0 = (long)blockIdx.x * rpb + rl; r0 < rows; r0 += 4 * stride) {
        uint4 uy[4];
        float2 g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long r = r0 + t * stride;
            uy[t] = make_uint4(0u, 0u, 0u, 0u); g[t] = make_float2(0.f, 0.f);
            if (r < rows) { uy[t] = *reinterpret_cast<const uint4*>(y + r * C + q * 8); g[t] = dlin[r]; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long r = r0 + t * stride;
            if (r >= rows) continue;
            const RChunk8 a = r_unpack8(uy[t]);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = k[j].x * a.v[j] + k[j].y;
                const float dzv = g[t].x * lc.wa[j] + g[t].y * lc.wb[j];
                const float gg = v > 0.f ? dzv : RBN_SLOPE * dzv;
                const float xh = (a.v[j] - k[j].z) * k[j].w;
                o[j] = ka[j] * (gg - k1[j] - xh * k2[j]);
            }
            *reinterpret_cast<uint4*>(dy + r * C + q * 8) = r_pack8(o);
        }
    }
}

static int check_dcu(const char* who, int R, int F, int T, int Cs, int Cr, int mode) {
    SEHIP_REQUIRE(R > 0 && F > 0 && T > 0, "%s: empty input", who);
    SEHIP_REQUIRE(Cs >= 8 && Cs <= 128 && (Cs & (Cs - 1)) == 0 && Cr >= 1 && Cr <= Cs, "%s: bad channel counts Cs=%d Cr=%d", who, Cs, Cr);
    SEHIP_REQUIRE(mode >= 0 && mode <= 2, "%s: masking mode must be 0(E) 1(C) 2(R)", who);
    return 0;
}

extern "C" int sehip_dcunet_pack_input(const float* spec, int R, int F, int T, void* out_bf16, void* stream) {
    SEHIP_REQUIRE(R > 0 && F > 0 && T > 0, "dcunet_pack_input: empty input");
    dcunet_pack_input_kernel<<<dim3(cdiv(T, DT), cdiv(F, DT), R), 256, 0, (hipStream_t)stream>>>((const float2*)spec, F, T, (unsigned*)out_bf16);
    SEHIP_CHECK_LAUNCH("dcunet_pack_input");
    return 0;
}

extern "C" int sehip_dcunet_mask_fwd(const void* z_bf16, const float* w_re, const float* w_im, const float* b_re, const float* b_im,
                                     const float* spec, int R, int F, int T, int Cs, int Cr, int mode, float* mask_ws, float* out,
                                     void* stream) {
    if (int e = check_dcu("dcunet_mask_fwd", R, F, T, Cs, Cr, mode)) return e;
    dcunet_mask_fwd_kernel<<<dim3(cdiv(T, DT), cdiv(F, DT), R), 256, 0, (hipStream_t)stream>>>(
        (const bf16_raw*)z_bf16, w_re, w_im, b_re, b_im, (const float2*)spec, F, T, Cs, Cr, mode, (float2*)mask_ws, (float2*)out, nullptr);
    SEHIP_CHECK_LAUNCH("dcunet_mask_fwd");
    return 0;
}

// the same from the last decoder's PRE-BatchNorm output y and that BatchNorm's coefficient records (sehip_rbn_finalize[_s]): the
// BatchNorm + LeakyReLU output is never stored (src/model/dcunet.py:40-50 decoder tail + :93-95, :131-159)
extern "C" int sehip_dcunet_mask_fwd_bn(const void* y_bf16, const float* coef, const float* w_re, const float* w_im, const float* b_re,
                                        const float* b_im, const float* spec, int R, int F, int T, int Cs, int Cr, int mode,
                                        float* mask_ws, float* out, void* stream) {
    if (int e = check_dcu("dcunet_mask_fwd_bn", R, F, T, Cs, Cr, mode)) return e;
    SEHIP_REQUIRE(coef != nullptr, "dcunet_mask_fwd_bn: missing BatchNorm coefficients");
    dcunet_mask_fwd_kernel<<<dim3(cdiv(T, DT), cdiv(F, DT), R), 256, 0, (hipStream_t)stream>>>(
        (const bf16_raw*)y_bf16, w_re, w_im, b_re, b_im, (const float2*)spec, F, T, Cs, Cr, mode, (float2*)mask_ws, (float2*)out,
        (const float4*)coef);
    SEHIP_CHECK_LAUNCH("dcunet_mask_fwd_bn");
    return 0;
}

// scratch of sehip_dcunet_tail_bwd (floats): [blocks][4][2 Cs] partial sums + [tiles][2] bias partials
extern "C" long sehip_dcunet_tail_scratch_floats(int R, int F, int T, int Cs) {
    const long rows = (long)R * T * F;
    return (long)rbn_stat_blocks(rows, 2 * Cs) * 4L * 2 * Cs + 2L * cdiv(T, DT) * cdiv(F, DT) * R;
}

// backward of mask + tanh + 1x1 conv + the last decoder's BatchNorm + LeakyReLU in one entry (see the kernels): d enhanced ->
// d(pre-BatchNorm output) dy, BatchNorm's weight / bias gradients (overwritten), the 1x1 conv's gradients gacc (accumulated: the
// caller zeroes).  mask_ws: in tanh(linear) of the forward pass, out d linear.  No tensor-sized intermediate is written.
extern "C" int sehip_dcunet_tail_bwd(const float* dout, const float* spec, float* mask_ws, const void* y_bf16, const float* coef,
                                     const float* w_re, const float* w_im, int R, int F, int T, int Cs, int Cr, int mode, float* scratch,
                                     float* gw_re, float* gb_re, float* gw_im, float* gb_im, float* bcoef, void* dy_bf16, float* gacc,
                                     void* stream) {
    if (int e = check_dcu("dcunet_tail_bwd", R, F, T, Cs, Cr, mode)) return e;
    SEHIP_REQUIRE(scratch && coef && bcoef && gacc, "dcunet_tail_bwd: missing buffer");
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)R * T * F;
    const int C = 2 * Cs, nblk = rbn_stat_blocks(rows, C), ntile = cdiv(T, DT) * cdiv(F, DT) * R;
    float* part = scratch;
    float* bpart = scratch + (size_t)nblk * 4 * C;
    dcunet_dlin_kernel<<<dim3(cdiv(T, DT), cdiv(F, DT), R), 256, 0, st>>>((const float2*)dout, (const float2*)spec, (float2*)mask_ws, F, T, mode, bpart);
    dcunet_tail_reduce_kernel<<<nblk, 256, 0, st>>>((const float2*)mask_ws, (const bf16_raw*)y_bf16, (const float4*)coef, w_re, w_im, rows, Cs, Cr, part);
    dcunet_tail_finalize_kernel<<<C, 64, 0, st>>>(part, nblk, bpart, ntile, (const float4*)coef, rows, Cs, Cr, gw_re, gb_re, gw_im, gb_im,
                                                   (float4*)bcoef, gacc);
    dcunet_tail_apply_kernel<<<rbn_apply_blocks(rows, C), 256, 0, st>>>((const float2*)mask_ws, (const bf16_raw*)y_bf16, (const float4*)coef,
                                                                         (const float4*)bcoef, w_re, w_im, rows, Cs, Cr, (bf16_raw*)dy_bf16);
    SEHIP_CHECK_LAUNCH("dcunet_tail_bwd");
    return 0;
}

extern "C" int sehip_dcunet_mask_bwd(const float* dout, const float* spec, const float* mask_ws, const void* z_bf16, const float* w_re,
                                     const float* w_im, int R, int F, int T, int Cs, int Cr, int mode, void* dz_bf16, float* gacc,
                                     void* stream) {
    if (int e = check_dcu("dcunet_mask_bwd", R, F, T, Cs, Cr, mode)) return e;
    float* part = nullptr;
    const int nrows = cdiv(T, DT) * cdiv(F, DT) * R;
    if (sehip_deterministic()) {      // fixed-order sums: one row of partial sums per workgroup (the per-stream pool of csrc/wgrad3.hip)
        part = sehip_wgrad_scratch((hipStream_t)stream, (size_t)nrows * 2 * (2 * Cs + 2) * sizeof(float));
        SEHIP_REQUIRE(part != nullptr, "dcunet_mask_bwd: no scratch for the deterministic schedule (inside a stream capture?)");
    }
    dcunet_mask_bwd_kernel<<<dim3(cdiv(T, DT), cdiv(F, DT), R), 256, 0, (hipStream_t)stream>>>(
        (const float2*)dout, (const float2*)spec, (const float2*)mask_ws, (const bf16_raw*)z_bf16, w_re, w_im, F, T, Cs, Cr, mode,
        (bf16_raw*)dz_bf16, gacc, part);
    if (part) dcunet_rows_reduce_kernel<<<cdiv(2 * Cs + 2, 256), 256, 0, (hipStream_t)stream>>>(part, nrows, Cs, Cr, gacc);
    SEHIP_CHECK_LAUNCH("dcunet_mask_bwd");
    return 0;
}
