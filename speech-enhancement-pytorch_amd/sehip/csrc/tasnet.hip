// ConvTasNet (src/model/conv_tasnet.py:34-487, shipped options: skip=False, gLN, non-causal, relu mask) -- everything that is not
// a 1x1 convolution (those are dense products on the implicit-GEMM engine of gemm.hip).  The network is memory-bound
// (3.2 GFLOP per 4-s clip against ~25 activation tensors of 0.4-0.8 MB per clip and block), so every kernel here is an HBM
// stream over channels-last bf16 activations [M][K][C] (row = one frame k of utterance m; a thread owns 8 channels = 16 bytes):
//
//   ctn_encoder_fwd        Conv1d(ac -> N, L, stride L/2, no bias) + ReLU (:157-176) fused with the channel-wise LayerNorm that
//                          follows it (:439-462): mixture_w (fp32, kept for the decoder) and cLN(mixture_w) (bf16) in one pass
//   ctn_gln_stats          per-utterance sum / sum of squares of PReLU(h) for the global LayerNorm (:465-487)
//   ctn_dwconv_fwd         n = gLN(PReLU(h1)) on the fly, depthwise dilated conv (groups = channels, 'same' zero padding, :366-379),
//                          stores h2 and accumulates the statistics of PReLU(h2) for the second gLN
//   ctn_gln_apply          u = gLN(PReLU(h2)) (the input of the pointwise 1x1 conv)
//   ctn_decoder_fwd        source_w = mixture_w * relu(mask logits) (:140, :196), basis_signals Linear(N -> ac*L) (:198) and
//                          overlap_and_add (:11-31) -> separated waveforms [M][C][ac][T]
//   ctn_*_bwd              the gradients of all of the above (two passes per LayerNorm: per-utterance sums, then apply)
//
//   gLN:  y = gamma (v - mu) / sqrt(var + 1e-8) + beta,  mu / var over (channels, frames) of ONE utterance;  v = PReLU(h; a)
//         dv = (gamma dy - mean(gamma dy) - xh mean(gamma dy xh)) / sigma;  dh = dv (h > 0 ? 1 : a);  da = sum dv h [h <= 0]
#include "common.h"
#include "det.h"

#define CTN_EPS 1e-8f

struct C8 { float v[8]; };
__device__ __forceinline__ C8 ld8(const bf16_raw* p) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    C8 c;
    c.v[0] = bf2f((bf16_raw)(u.x & 0xffff)); c.v[1] = bf2f((bf16_raw)(u.x >> 16));
    c.v[2] = bf2f((bf16_raw)(u.y & 0xffff)); c.v[3] = bf2f((bf16_raw)(u.y >> 16));
    c.v[4] = bf2f((bf16_raw)(u.z & 0xffff)); c.v[5] = bf2f((bf16_raw)(u.z >> 16));
    c.v[6] = bf2f((bf16_raw)(u.w & 0xffff)); c.v[7] = bf2f((bf16_raw)(u.w >> 16));
    return c;
}
__device__ __forceinline__ void st8(bf16_raw* p, const float* v) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ uint4 ld8raw(const bf16_raw* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ C8 unpack8(const uint4 u) {
    C8 c;
    c.v[0] = bf2f((bf16_raw)(u.x & 0xffff)); c.v[1] = bf2f((bf16_raw)(u.x >> 16));
    c.v[2] = bf2f((bf16_raw)(u.y & 0xffff)); c.v[3] = bf2f((bf16_raw)(u.y >> 16));
    c.v[4] = bf2f((bf16_raw)(u.z & 0xffff)); c.v[5] = bf2f((bf16_raw)(u.z >> 16));
    c.v[6] = bf2f((bf16_raw)(u.w & 0xffff)); c.v[7] = bf2f((bf16_raw)(u.w >> 16));
    return c;
}
__device__ __forceinline__ C8 zero8() { C8 c; for (int j = 0; j < 8; ++j) c.v[j] = 0.f; return c; }
__device__ __forceinline__ float prelu(float h, float a) { return h > 0.f ? h : a * h; }

// mean and 1/sigma of an utterance from its (sum, sumsq) record
__device__ __forceinline__ void gln_moments(const double* __restrict__ st, int m, long n, float& mu, float& rs) {
    const double s = st[2 * m], q = st[2 * m + 1];
    const double mean = s / (double)n;
    double var = q / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    mu = (float)mean;
    rs = 1.f / sqrtf((float)var + CTN_EPS);
}

// ------------------------------------------------------------------------------------------------------------------
// encoder + cLN.  One wave per frame; lane owns channels lane, lane + 64, ... (N <= 256); U^T in LDS.
// ------------------------------------------------------------------------------------------------------------------
#define ENC_MAXC 8        // the widest instantiation of the wave-per-frame kernels below
template <int MC>        // channels per lane: N <= 64 MC (4: N <= 256, 8: N <= 512)
__global__ __launch_bounds__(256) void ctn_encoder_fwd_kernel(const float* __restrict__ wav, const float* __restrict__ U /*[N][ac*L]*/,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int M,
                                                              int ac, int T, int K, int N, int L, float* __restrict__ w,
                                                              bf16_raw* __restrict__ cln) {
    extern __shared__ float sU[];                    // [ac*L][N]
    const int AL = ac * L;
    for (int i = threadIdx.x; i < N * AL; i += 256) { const int n = i / AL, l = i - n * AL; sU[l * N + n] = U[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int step = L / 2;
    const long frames = (long)M * K;
    for (long fr = (long)blockIdx.x * 4 + wave; fr < frames; fr += (long)gridDim.x * 4) {
        const int m = (int)(fr / K), k = (int)(fr - (long)m * K);
        float acc[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) acc[i] = 0.f;
        for (int a = 0; a < ac; ++a) {
            const float* x = wav + ((long)m * ac + a) * T + (long)k * step;
            for (int l = 0; l < L; ++l) {
                const float xv = x[l];
#pragma unroll
                for (int i = 0; i < MC; ++i) {
                    const int n = lane + 64 * i;
                    if (n < N) acc[i] += xv * sU[(a * L + l) * N + n];
                }
            }
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            acc[i] = acc[i] > 0.f ? acc[i] : 0.f;
            if (lane + 64 * i < N) { s += acc[i]; q += acc[i] * acc[i]; }
        }
        s = wave_sum(s); q = wave_sum(q);
        const float mean = s / N;
        float var = q / N - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rs = 1.f / sqrtf(var + CTN_EPS);
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int n = lane + 64 * i;
            if (n < N) {
                w[fr * N + n] = acc[i];
                cln[fr * N + n] = f2bf(gamma[n] * (acc[i] - mean) * rs + beta[n]);
            }
        }
    }
}

// The same on the MFMA pipe (one audio channel, N = 128, L <= 64): 16 frames per tile,
//   D[n][frame] = sum_l U[n][l] x[frame][l]   (U = A operand, in registers for the whole launch; x = B operand, 8 consecutive
// samples per lane straight from the waveform), both split into bf16 high + low parts (three MFMAs, ~2^-16: `w` multiplies the
// mask at the decoder, it stays at fp32 accuracy).  A lane ends up with 32 of its frame's 128 channels (4 consecutive ones per
// 16-channel tile); the channel-wise LayerNorm sums meet the other three lanes of the frame through two xor-shuffles.  The
// wave-per-frame kernel above walks 40 taps with a dependent waveform load and two LDS reads each: 84 us against 43 MB of traffic.
template <int KS>
__global__ __launch_bounds__(256) void ctn_encoder_fwd_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ U /*[128][L]*/,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int M,
                                                                   int T, int K, int L, float* __restrict__ w, bf16_raw* __restrict__ cln) {
    constexpr int N = 128, TNN = 8;
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    auto split8 = [](const float (&x)[8], bf16x8& hi, bf16x8& lo) {
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_raw h0 = f2bf(x[2 * i]), h1 = f2bf(x[2 * i + 1]);
            h[i] = (unsigned)h0 | ((unsigned)h1 << 16);
            l[i] = pack_bf2(x[2 * i] - bf2f(h0), x[2 * i + 1] - bf2f(h1));
        }
        hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
        lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    };
    bf16x8 uhi[TNN][KS], ulo[TNN][KS];
#pragma unroll
    for (int tn = 0; tn < TNN; ++tn)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int l0 = 32 * ks + 8 * g;
            if (l0 + 8 <= L && (L & 3) == 0) {          // (rows of U are 16-byte aligned when L is a multiple of 4)
                const float4 v0 = *reinterpret_cast<const float4*>(U + (size_t)(16 * tn + c16) * L + l0);
                const float4 v1 = *reinterpret_cast<const float4*>(U + (size_t)(16 * tn + c16) * L + l0 + 4);
                x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (l0 + j < L) x[j] = U[(size_t)(16 * tn + c16) * L + l0 + j];
            }
            split8(x, uhi[tn][ks], ulo[tn][ks]);
        }
    float4 gm[TNN], bt[TNN];
#pragma unroll
    for (int tn = 0; tn < TNN; ++tn) {
        gm[tn] = *reinterpret_cast<const float4*>(gamma + 16 * tn + 4 * g);
        bt[tn] = *reinterpret_cast<const float4*>(beta + 16 * tn + 4 * g);
    }
    const int step = L / 2;
    const long frames = (long)M * K;
    const long ntiles = (frames + 15) / 16;
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const long fr = tile * 16 + c16;
        const bool rok = fr < frames;
        const long frc = rok ? fr : 0;
        const int m = (int)(frc / K), k = (int)(frc - (long)m * K);
        const float* x0 = wav + (long)m * T + (long)k * step;
        bf16x8 xhi[KS], xlo[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int l = 32 * ks + 8 * g + j;
                x[j] = (rok && l < L) ? x0[l] : 0.f;
            }
            split8(x, xhi[ks], xlo[ks]);
        }
        f32x4 acc[TNN];
#pragma unroll
        for (int tn = 0; tn < TNN; ++tn) {
            acc[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uhi[tn][ks], xhi[ks], acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ulo[tn][ks], xhi[ks], acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uhi[tn][ks], xlo[ks], acc[tn], 0, 0, 0);
            }
        }
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int tn = 0; tn < TNN; ++tn)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[tn][e] > 0.f ? acc[tn][e] : 0.f;
                acc[tn][e] = v; s += v; q += v * v;
            }
        s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
        s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
        const float mean = s / N;
        float var = q / N - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rs = 1.f / sqrtf(var + CTN_EPS);
        if (rok) {
#pragma unroll
            for (int tn = 0; tn < TNN; ++tn) {
                const f32x4 v = acc[tn];
                *reinterpret_cast<float4*>(w + fr * N + 16 * tn + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<uint2*>(cln + fr * N + 16 * tn + 4 * g) =
                    make_uint2(pack_bf2(gm[tn].x * (v[0] - mean) * rs + bt[tn].x, gm[tn].y * (v[1] - mean) * rs + bt[tn].y),
                               pack_bf2(gm[tn].z * (v[2] - mean) * rs + bt[tn].z, gm[tn].w * (v[3] - mean) * rs + bt[tn].w));
            }
        }
    }
}

// backward of cLN + ReLU + encoder conv:  dw = dw_dec + cLN'(dcln);  dpre = dw [w > 0];  dU[n][l] += dpre x[l]
// gacc: dU [N][ac*L] | dgamma [N] | dbeta [N]   (fp32, atomics; caller zeroes)
template <int MC>        // channels per lane: N <= 64 MC (4: N <= 256, 8: N <= 512)
__global__ __launch_bounds__(256) void ctn_encoder_bwd_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                              const bf16_raw* __restrict__ dcln, const float* __restrict__ dw_dec,
                                                              const float* __restrict__ gamma, int M, int ac, int T, int K, int N, int L,
                                                              int frames_per_wave, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int step = L / 2, AL = ac * L;
    const long frames = (long)M * K;
    const long f0 = ((long)blockIdx.x * 4 + wave) * frames_per_wave;
    float dg[MC], db[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) { dg[i] = 0.f; db[i] = 0.f; }
    // dU accumulators: this lane's channels x taps, kept in LDS rows private to the wave (MC * AL floats per lane is too
    // many registers for L = 40): [wave][l][n]
    extern __shared__ float sdU[];
    float* mine = sdU + (size_t)wave * AL * N;
    for (int i = lane; i < AL * N; i += 64) mine[i] = 0.f;
    float gam[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) gam[i] = lane + 64 * i < N ? gamma[lane + 64 * i] : 0.f;
    for (long fr = f0; fr < f0 + frames_per_wave && fr < frames; ++fr) {
        const int m = (int)(fr / K), k = (int)(fr - (long)m * K);
        float wv[MC], dy[MC];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int n = lane + 64 * i;
            wv[i] = n < N ? w[fr * N + n] : 0.f;
            dy[i] = n < N ? bf2f(dcln[fr * N + n]) : 0.f;
            s += wv[i]; q += wv[i] * wv[i];
        }
        s = wave_sum(s); q = wave_sum(q);
        const float mean = s / N;
        float var = q / N - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rs = 1.f / sqrtf(var + CTN_EPS);
        float s1 = 0.f, s2 = 0.f, xh[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            xh[i] = (wv[i] - mean) * rs;
            if (lane + 64 * i < N) { s1 += gam[i] * dy[i]; s2 += gam[i] * dy[i] * xh[i]; dg[i] += dy[i] * xh[i]; db[i] += dy[i]; }
        }
        s1 = wave_sum(s1) / N; s2 = wave_sum(s2) / N;
        float dpre[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int n = lane + 64 * i;
            float dwv = n < N ? (gam[i] * dy[i] - s1 - xh[i] * s2) * rs + dw_dec[fr * N + n] : 0.f;
            dpre[i] = wv[i] > 0.f ? dwv : 0.f;
        }
        for (int a = 0; a < ac; ++a) {
            const float* x = wav + ((long)m * ac + a) * T + (long)k * step;
            for (int l = 0; l < L; ++l) {
                const float xv = x[l];
#pragma unroll
                for (int i = 0; i < MC; ++i) {
                    const int n = lane + 64 * i;
                    if (n < N) mine[(a * L + l) * N + n] += dpre[i] * xv;
                }
            }
        }
    }
    // One row of partial sums per workgroup in gacc layout (dU [N][AL] | dgamma [N] | dbeta [N]); ctn_colsum_kernel adds the
    // rows.  (Flushing every wave with atomics was 2 048 waves x 5 120 addresses: 1.2 ms of a 1.24-ms kernel.)
    float* sgb = sdU + (size_t)4 * AL * N;             // [2][N] behind the four waves' dU images
    for (int i = threadIdx.x; i < 2 * N; i += 256) sgb[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {          // the four waves one after the other: a fixed order (LDS atomics have the hardware's)
        if (wave == turn) {
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const int n = lane + 64 * i;
                if (n < N) { sgb[n] += dg[i]; sgb[N + n] += db[i]; }
            }
        }
        __syncthreads();
    }
    float* row = part + (size_t)blockIdx.x * (AL * N + 2 * N);
    for (int i = threadIdx.x; i < AL * N; i += 256) {
        const int l = i / N, n = i - l * N;
        row[n * AL + l] = sdU[i] + sdU[AL * N + i] + sdU[2 * AL * N + i] + sdU[3 * AL * N + i];
    }
    for (int i = threadIdx.x; i < 2 * N; i += 256) row[AL * N + i] = sgb[i];
}

// The same pass with the dU accumulators in REGISTERS (frame length AL = ac*L known at compile time, AL <= 64): lane n keeps
// dU[n][0..AL) of its NC channels; the frame's AL input samples sit one per lane and reach the FMAs through v_readlane.
// The LDS version above pays 2 LDS operations per multiply-add (0.48 ms at C4); this one none.
template <int AL, int NC>
__global__ __launch_bounds__(256) void ctn_encoder_bwd_reg_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                                  const bf16_raw* __restrict__ dcln, const float* __restrict__ dw_dec,
                                                                  const float* __restrict__ gamma, int M, int ac, int T, int K, int N, int L,
                                                                  int frames_per_wave, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int step = L / 2;
    const long frames = (long)M * K;
    const long f0 = ((long)blockIdx.x * 4 + wave) * frames_per_wave;
    float dg[NC], db[NC], gam[NC], acc[NC][AL];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        dg[i] = 0.f; db[i] = 0.f;
        gam[i] = lane + 64 * i < N ? gamma[lane + 64 * i] : 0.f;
#pragma unroll
        for (int l = 0; l < AL; ++l) acc[i][l] = 0.f;
    }
    const int xa = lane / L, xl = lane - xa * L;          // this lane's sample of the frame: channel xa, tap xl (lane < AL)
    for (long fr = f0; fr < f0 + frames_per_wave && fr < frames; ++fr) {
        const int m = (int)(fr / K), k = (int)(fr - (long)m * K);
        const float xv = lane < AL ? wav[((long)m * ac + xa) * T + (long)k * step + xl] : 0.f;
        float wv[NC], dy[NC];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int n = lane + 64 * i;
            wv[i] = n < N ? w[fr * N + n] : 0.f;
            dy[i] = n < N ? bf2f(dcln[fr * N + n]) : 0.f;
            s += wv[i]; q += wv[i] * wv[i];
        }
        s = wave_sum(s); q = wave_sum(q);
        const float mean = s / N;
        float var = q / N - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rs = 1.f / sqrtf(var + CTN_EPS);
        float s1 = 0.f, s2 = 0.f, xh[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            xh[i] = (wv[i] - mean) * rs;
            if (lane + 64 * i < N) { s1 += gam[i] * dy[i]; s2 += gam[i] * dy[i] * xh[i]; dg[i] += dy[i] * xh[i]; db[i] += dy[i]; }
        }
        s1 = wave_sum(s1) / N; s2 = wave_sum(s2) / N;
        float dpre[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int n = lane + 64 * i;
            const float dwv = n < N ? (gam[i] * dy[i] - s1 - xh[i] * s2) * rs + dw_dec[fr * N + n] : 0.f;
            dpre[i] = wv[i] > 0.f ? dwv : 0.f;
        }
#pragma unroll
        for (int l = 0; l < AL; ++l) {
            const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), l));
#pragma unroll
            for (int i = 0; i < NC; ++i) acc[i][l] += dpre[i] * x;
        }
    }
    // the four waves' sums meet in LDS; one row per workgroup in gacc layout (dU [N][AL] | dgamma [N] | dbeta [N])
    extern __shared__ float srow[];                       // [N*AL + 2N]
    const int ncols = N * AL + 2 * N;
    for (int i = threadIdx.x; i < ncols; i += 256) srow[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {          // the four waves one after the other: a fixed order (LDS atomics have the hardware's)
        if (wave == turn) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int n = lane + 64 * i;
                if (n < N) {
#pragma unroll
                    for (int l = 0; l < AL; ++l) srow[n * AL + l] += acc[i][l];
                    srow[N * AL + n] += dg[i];
                    srow[N * AL + N + n] += db[i];
                }
            }
        }
        __syncthreads();
    }
    float* row = part + (size_t)blockIdx.x * ncols;
    for (int i = threadIdx.x; i < ncols; i += 256) row[i] = srow[i];
}

// The same pass on the MFMA pipe (one audio channel, N = 128, L <= 16 TL): tiles of 32 frames,
//   dU[n][l] += sum_frames dpre[frame][n] x[frame][l]   as   D[n][l] = A[n][frame] B[frame][l].
// Lane (c16, g) does the elementwise part (cLN backward + ReLU mask) for channels 16 tn + c16 (tn < 8) of frames 8 g .. 8 g + 7 --
// exactly the 8 consecutive k values of the MFMA's A operand row n, so dpre goes from registers straight into the MFMA; the
// per-frame sums over the 128 channels are four xor-shuffles over the 16 lanes that share g.  B: 8 samples x[frame][l] per lane,
// strided loads from the waveform, split into bf16 high + low parts (dpre is rounded to bf16 once: a weight gradient).  One
// partial row per workgroup as above.
// (The register kernel above walks the frames one by one, 40 v_readlane + 80 FMA steps each: 195 us at the C4 shape.)
template <int TL, int ABL = 0>
__global__ __launch_bounds__(256, 2) void ctn_encoder_bwd_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                                      const bf16_raw* __restrict__ dcln, const float* __restrict__ dw_dec,
                                                                      const float* __restrict__ gamma, int M, int T, int K, int L,
                                                                      float* __restrict__ part) {
    constexpr int N = 128, TNN = 8;
    // per wave: the A operand as it is built, [tn][lane][8 frames] bf16, and the B operand's samples, [tl][lane][8 frames] fp32 -- a
    // lane reads back exactly what it wrote (the frame loop is a real loop: unrolled, its 216 loads per tile took 512 registers)
    extern __shared__ __attribute__((aligned(16))) unsigned char ebw_smem[];
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    constexpr int WBYTES = TNN * 64 * 16 + TL * 64 * 32;
    bf16_raw* sdp = reinterpret_cast<bf16_raw*>(ebw_smem + wave * WBYTES);
    float* sxs = reinterpret_cast<float*>(ebw_smem + wave * WBYTES + TNN * 64 * 16);
    float* srow = reinterpret_cast<float*>(ebw_smem + 4 * WBYTES);          // [N*L + 2N]
    auto split8 = [](const float (&x)[8], bf16x8& hi, bf16x8& lo) {
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_raw h0 = f2bf(x[2 * i]), h1 = f2bf(x[2 * i + 1]);
            h[i] = (unsigned)h0 | ((unsigned)h1 << 16);
            l[i] = pack_bf2(x[2 * i] - bf2f(h0), x[2 * i + 1] - bf2f(h1));
        }
        hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
        lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    };
    auto sum16 = [](float v) {          // over the 16 lanes that share g
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        return v;
    };
    float gam[TNN], dgl[TNN], dbl[TNN];
#pragma unroll
    for (int tn = 0; tn < TNN; ++tn) { gam[tn] = gamma[16 * tn + c16]; dgl[tn] = 0.f; dbl[tn] = 0.f; }
    f32x4 acc[TNN][TL];
#pragma unroll
    for (int tn = 0; tn < TNN; ++tn)
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) acc[tn][tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int step = L / 2;
    const long frames = (long)M * K;
    const long ntiles = (frames + 31) / 32;
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    for (long tile = wave_id; tile < ((ABL & 2) ? 0 : ntiles); tile += nwaves) {
        const long f0 = tile * 32 + 8 * g;
#pragma unroll 2
        for (int j = 0; j < 8; ++j) {
            const long f = f0 + j;
            const bool ok = f < frames;
            const long fc = ok ? f : 0;
            float wv[TNN], dy[TNN], dd[TNN];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int tn = 0; tn < TNN; ++tn) {
                wv[tn] = w[fc * N + 16 * tn + c16];
                dy[tn] = bf2f(dcln[fc * N + 16 * tn + c16]);
                dd[tn] = dw_dec[fc * N + 16 * tn + c16];
                s += wv[tn]; q += wv[tn] * wv[tn];
            }
            const int m = (int)(fc / K), k = (int)(fc - (long)m * K);
            const float* x0 = wav + (long)m * T + (long)k * step;
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) {
                const int l = 16 * tl + c16;
                sxs[(tl * 64 + lane) * 8 + j] = (ok && l < L) ? x0[l] : 0.f;
            }
            s = sum16(s); q = sum16(q);
            const float mean = s / N;
            float var = q / N - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float rs = 1.f / sqrtf(var + CTN_EPS);
            float s1 = 0.f, s2 = 0.f, xh[TNN];
#pragma unroll
            for (int tn = 0; tn < TNN; ++tn) {
                xh[tn] = (wv[tn] - mean) * rs;
                s1 += gam[tn] * dy[tn]; s2 += gam[tn] * dy[tn] * xh[tn];
                if (ok) { dgl[tn] += dy[tn] * xh[tn]; dbl[tn] += dy[tn]; }
            }
            s1 = sum16(s1) / N; s2 = sum16(s2) / N;
#pragma unroll
            for (int tn = 0; tn < TNN; ++tn) {
                const float dwv = (gam[tn] * dy[tn] - s1 - xh[tn] * s2) * rs + dd[tn];
                sdp[(tn * 64 + lane) * 8 + j] = f2bf((ok && wv[tn] > 0.f) ? dwv : 0.f);
            }
        }
        // (a lane reads its own LDS words: the LDS queue of a wave is in order, no barrier)
        bf16x8 xhi[TL], xlo[TL];
#pragma unroll
        for (int tl = 0; tl < TL; ++tl) {
            const float4 a = *reinterpret_cast<const float4*>(&sxs[(tl * 64 + lane) * 8]);
            const float4 b = *reinterpret_cast<const float4*>(&sxs[(tl * 64 + lane) * 8 + 4]);
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            split8(x, xhi[tl], xlo[tl]);
        }
#pragma unroll
        for (int tn = 0; tn < TNN; ++tn) {
            const bf16x8 a8 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&sdp[(tn * 64 + lane) * 8]));
#pragma unroll
            for (int tl = 0; tl < TL; ++tl) {
                acc[tn][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, xhi[tl], acc[tn][tl], 0, 0, 0);
                acc[tn][tl] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, xlo[tl], acc[tn][tl], 0, 0, 0);
            }
        }
    }
    // the four waves' sums meet in LDS, one wave after the other with plain read-add-write (inside a wave every address is touched
    // by one lane).  With ds_add_f32 from all four waves at once the 112 atomics per lane took 78 of this kernel's 98 us (~200
    // cycles per wave instruction); one row per workgroup in gacc layout (dU [N][L] | dgamma [N] | dbeta [N])
    const int ncols = N * L + 2 * N;
    for (int i = threadIdx.x; i < ncols; i += 256) srow[i] = 0.f;
#pragma unroll
    for (int tn = 0; tn < TNN; ++tn) {          // dgamma / dbeta: the four g lanes of a wave hold different frames of the same channel
        dgl[tn] += __shfl_xor(dgl[tn], 16, 64); dgl[tn] += __shfl_xor(dgl[tn], 32, 64);
        dbl[tn] += __shfl_xor(dbl[tn], 16, 64); dbl[tn] += __shfl_xor(dbl[tn], 32, 64);
    }
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn && !(ABL & 1)) {
#pragma unroll
            for (int tn = 0; tn < TNN; ++tn) {
#pragma unroll
                for (int tl = 0; tl < TL; ++tl) {
                    const int l = 16 * tl + c16;          // D row = channel 16 tn + 4 g + e, D column = sample l
                    if (l < L) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) srow[(16 * tn + 4 * g + e) * L + l] += acc[tn][tl][e];
                    }
                }
                if (g == 0) {
                    srow[N * L + 16 * tn + c16] += dgl[tn];
                    srow[N * L + N + 16 * tn + c16] += dbl[tn];
                }
            }
        }
        __syncthreads();
    }
    float* row = part + (size_t)blockIdx.x * ncols;
    for (int i = threadIdx.x; i < ncols; i += 256) row[i] = srow[i];
}

// ------------------------------------------------------------------------------------------------------------------
// global LayerNorm pieces.  Rows of utterance m: [m*K, (m+1)*K).  grid = (blocks per utterance, M).
// ------------------------------------------------------------------------------------------------------------------
// dst[0..1] += the workgroup's (s, q).  Default schedule: two double atomics per workgroup; deterministic schedule (dc.part != NULL):
// the workgroups of utterance blockIdx.y leave their pairs in slots and the wrapper's sehip_det_finish adds them in a fixed order
// (csrc/det.h).
__device__ __forceinline__ void block_add2_double(float s, float q, double* dst, const DetCtx dc) {
    __shared__ float red[2][4];
    __shared__ double pair[2];
    s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        pair[0] = (double)red[0][0] + red[0][1] + red[0][2] + red[0][3];
        pair[1] = (double)red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    __syncthreads();
    det_group_add(pair, 2, dst, dc, blockIdx.y, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void ctn_gln_stats_kernel(const bf16_raw* __restrict__ h, const float* __restrict__ slope, int K, int C,
                                                            double* __restrict__ stats, const DetCtx dc) {
    const int m = blockIdx.y, nq = C >> 3;
    const float a = slope[0];
    const long pieces = (long)K * nq;
    const bf16_raw* base = h + (long)m * K * C;
    float s = 0.f, q = 0.f;
    // (three pieces in flight per thread measured the same 11.2 us per launch: 26 MB at the launch floor)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (long)gridDim.x * 256) {
        const C8 x = ld8(base + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = prelu(x.v[j], a); s += v; q += v * v; }
    }
    block_add2_double(s, q, stats + 2 * m, dc);
}

// Thread layout of the frame-streaming kernels below: a thread owns ONE piece of 8 channels (q = tid % nq) for all its
// frames, so gamma / beta / the depthwise taps of those channels are loaded once and stay in registers (fetched per piece
// they were 40 four-byte loads beside 3-8 sixteen-byte ones, and the kernels ran at the address unit's pace: 90-260 us for
// 26-MB tensors); rows t = bx*rpb + tid/nq, stepping by gridDim.x*rpb (rpb = 256/nq rows per block pass).
struct PieceMap { int q, c0, rsub, rpb; bool active; };
__device__ __forceinline__ PieceMap piece_map(int nq) {
    PieceMap p;
    p.rpb = 256 / nq;
    p.q = threadIdx.x % nq;
    p.rsub = threadIdx.x / nq;
    p.c0 = p.q * 8;
    p.active = p.rsub < p.rpb;
    return p;
}
__device__ __forceinline__ void ld8f(const float* __restrict__ p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// h2[t][c] = sum_j Wd[c][j] n1[t + (j - P/2) d][c],  n1 = gLN1(PReLU(h1)) (zero outside [0, K)); stats2 += PReLU(h2; a2)
template <int P>
__global__ __launch_bounds__(256) void ctn_dwconv_fwd_kernel(const bf16_raw* __restrict__ h1, const float* __restrict__ slope1,
                                                             const double* __restrict__ stats1, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ Wd /*[C][P]*/,
                                                             int dil, const float* __restrict__ slope2, int K, int C,
                                                             bf16_raw* __restrict__ h2, double* __restrict__ stats2, const DetCtx dc) {
    const int m = blockIdx.y, nq = C >> 3;
    const PieceMap pm = piece_map(nq);
    const float a1 = slope1[0], a2 = slope2[0];
    float mu, rs;
    gln_moments(stats1, m, (long)K * C, mu, rs);
    const bf16_raw* base = h1 + (long)m * K * C + pm.c0;
    bf16_raw* out = h2 + (long)m * K * C + pm.c0;
    float s = 0.f, q = 0.f;
    if (pm.active) {
        // n1 = gs * prelu(x) + bs  with  gs = gamma rs,  bs = beta - gamma mu rs
        float gs[8], bs[8], wd[P][8];
        ld8f(gamma + pm.c0, gs); ld8f(beta + pm.c0, bs);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gs[j] *= rs; bs[j] -= gs[j] * mu;
#pragma unroll
            for (int p = 0; p < P; ++p) wd[p][j] = Wd[(pm.c0 + j) * P + p];
        }
        // (two output rows -- six loads -- in flight per thread measured 22.8 us per launch against 21.6: one row)
        for (int t = blockIdx.x * pm.rpb + pm.rsub; t < K; t += gridDim.x * pm.rpb) {
            C8 x[P];
            bool ok[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int tt = t + (p - P / 2) * dil;
                ok[p] = tt >= 0 && tt < K;
                x[p] = ld8(base + (long)(ok[p] ? tt : t) * C);          // all loads in flight together (the centre row stands in)
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p)
                if (ok[p]) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += wd[p][j] * (gs[j] * prelu(x[p].v[j], a1) + bs[j]);
                }
            st8(out + (long)t * C, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float v = prelu(bf2f(f2bf(o[j])), a2); s += v; q += v * v; }   // statistics of what is stored
        }
    }
    block_add2_double(s, q, stats2 + 2 * m, dc);
}

__global__ __launch_bounds__(256) void ctn_gln_apply_kernel(const bf16_raw* __restrict__ h, const float* __restrict__ slope,
                                                            const double* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int K, int C, bf16_raw* __restrict__ u) {
    const int m = blockIdx.y, nq = C >> 3;
    const PieceMap pm = piece_map(nq);
    if (!pm.active) return;
    const float a = slope[0];
    float mu, rs;
    gln_moments(stats, m, (long)K * C, mu, rs);
    const bf16_raw* base = h + (long)m * K * C + pm.c0;
    bf16_raw* out = u + (long)m * K * C + pm.c0;
    float gs[8], bs[8];
    ld8f(gamma + pm.c0, gs); ld8f(beta + pm.c0, bs);
#pragma unroll
    for (int j = 0; j < 8; ++j) { gs[j] *= rs; bs[j] -= gs[j] * mu; }
    for (int t = blockIdx.x * pm.rpb + pm.rsub; t < K; t += gridDim.x * pm.rpb) {
        const C8 x = ld8(base + (long)t * C);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = gs[j] * prelu(x.v[j], a) + bs[j];
        st8(out + (long)t * C, o);
    }
}

// Backward of y = gLN(PReLU(h)).  The incoming gradient dy is either read directly (DW = false: du, the gradient of the
// pointwise conv's input) or is the transposed depthwise conv of dh2 (DW = true: dy[t][c] = sum_p Wd[c][p] dh2[t - (p - P/2) d][c]).
// pass 1 (reduce): per utterance S1 = sum gamma dy, S2 = sum gamma dy xh (double atomics into sums[2m..]); per channel
//                  dgamma += dy xh, dbeta += dy [, dWd[c][p] += dh2[t][c] n[t + (p - P/2) d][c]]: every block writes ONE row of
//                  per-channel partials (gch layout) to `part`, ctn_colsum_kernel adds the rows into gch.  (A first version
//                  flushed each block with fp32 atomics: 1 600 blocks x 1 280 addresses, 261 us for a 26-MB tensor.)
// pass 2 (apply) : dh = ((gamma dy - S1/n - xh S2/n) / sigma) (h > 0 ? 1 : a);  dslope += sum dv h [h <= 0]
// gch layout: dgamma [C] | dbeta [C] | dWd [C][P] (DW only)
template <int P, bool DW>
__global__ __launch_bounds__(256) void ctn_gln_bwd_reduce_kernel(const bf16_raw* __restrict__ g, const bf16_raw* __restrict__ h,
                                                                 const float* __restrict__ slope, const double* __restrict__ stats,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ Wd, int dil, int K, int C,
                                                                 double* __restrict__ sums, float* __restrict__ part, const DetCtx dc) {
    extern __shared__ float lds[];       // per-channel partials of the block in gch layout: [(2 + (DW ? P : 0)) * C]
    const int m = blockIdx.y, nq = C >> 3;
    const PieceMap pm = piece_map(nq);
    constexpr int NV = 2 + (DW ? P : 0);
    for (int i = threadIdx.x; i < NV * C; i += 256) lds[i] = 0.f;
    __syncthreads();
    const float a = slope[0];
    float mu, rs;
    gln_moments(stats, m, (long)K * C, mu, rs);
    const bf16_raw* gb = g + (long)m * K * C + pm.c0;
    const bf16_raw* hb = h + (long)m * K * C + pm.c0;
    float s1 = 0.f, s2 = 0.f;
    float qa = 0.f, qb = 0.f, qc = 0.f;      // slope gradient, see below: sums over the elements with h <= 0 of gamma dy h, h, xh h
    float dg[8], db[8], dw[DW ? P : 1][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        dg[j] = 0.f; db[j] = 0.f;
#pragma unroll
        for (int p = 0; p < (DW ? P : 1); ++p) dw[p][j] = 0.f;
    }
    if (pm.active) {
        float gm[8], bt[8], wd[DW ? P : 1][8];
        ld8f(gamma + pm.c0, gm); ld8f(beta + pm.c0, bt);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int p = 0; p < (DW ? P : 1); ++p) wd[p][j] = DW ? Wd[(pm.c0 + j) * P + p] : 0.f;
        }
        // Rows in flight (round 6): a workgroup walks ~12 trips of one row per thread, each a dependent round trip to memory with 2-4
        // loads behind it -- 3.0 TB/s for the DW = false pass.  U rows are requested before the first is used (more workgroups instead
        // cost the apply pass, which adds their partial rows: SEHIP_CTN_RBLOCKS).
        constexpr int U = DW ? 1 : 4;      // (measured: DW = false 24.3 -> 19.0 us per launch; DW = true 30.2 -> 30.1 with two rows: one)
        struct Row { uint4 x; uint4 gq[DW ? P : 1]; };        // (raw 16-byte pieces: un-packed when the row is used)
        auto load_row = [&](int t) {
            Row r;
            r.x = ld8raw(hb + (long)t * C);
            if (!DW) {
                r.gq[0] = ld8raw(gb + (long)t * C);
            } else {
                // rows t - d, t, t + d of dh2 serve BOTH sums (it was 3 more rows of h and the normalisation three times per element):
                // dy[t] = sum_p' Wd[p'] dh2[t - (p' - P/2) d], and dWd[p'] = sum_t dh2[t - (p' - P/2) d] n[t] -- the same pairs
                // (t - (p' - P/2) d, t) as sum_t dh2[t] n[t + (p' - P/2) d], indexed by the row that holds n
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int tt = t + (p - P / 2) * dil;
                    r.gq[p] = ld8raw(gb + (long)((tt >= 0 && tt < K) ? tt : t) * C);
                }
            }
            return r;
        };
        auto do_row = [&](const Row& r, int t) {
            const C8 x = unpack8(r.x);
            C8 dy;
            if (!DW) {
                dy = unpack8(r.gq[0]);
            } else {
                float nt[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) nt[j] = gm[j] * ((prelu(x.v[j], a) - mu) * rs) + bt[j];
                dy = zero8();
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int tt = t + (p - P / 2) * dil;      // the row t + (p - P/2) d is t - (p' - P/2) d for tap p' = P - 1 - p
                    if (tt >= 0 && tt < K) {
                        const C8 gp = unpack8(r.gq[p]);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            dy.v[j] += wd[P - 1 - p][j] * gp.v[j];
                            dw[P - 1 - p][j] += gp.v[j] * nt[j];
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (prelu(x.v[j], a) - mu) * rs;
                const float gd = gm[j] * dy.v[j];
                s1 += gd; s2 += gd * xh;
                dg[j] += dy.v[j] * xh; db[j] += dy.v[j];
                if (!(x.v[j] > 0.f)) { qa += gd * x.v[j]; qb += x.v[j]; qc += xh * x.v[j]; }
            }
        };
        const int stride = gridDim.x * pm.rpb;
        int t = blockIdx.x * pm.rpb + pm.rsub;
        for (; t + (U - 1) * stride < K; t += U * stride) {
            Row r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = load_row(t + u * stride);
#pragma unroll
            for (int u = 0; u < U; ++u) do_row(r[u], t + u * stride);
        }
        for (; t < K; t += stride) do_row(load_row(t), t);
    }
    // per-channel partials of the block.  nq a power of two <= 64: the lanes of a wave that hold the same channels (nq apart) meet by
    // xor-shuffles, then the four waves add their words one after the other with plain read-add-write (ds_add_f32 from all lanes
    // at once: 8 lanes per word, ~200 cycles per wave instruction, 40 of them); otherwise LDS atomics
    if ((nq & (nq - 1)) == 0 && nq <= 64) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            for (int o = nq; o < 64; o <<= 1) {
                dg[j] += __shfl_xor(dg[j], o, 64); db[j] += __shfl_xor(db[j], o, 64);
                if (DW) {
#pragma unroll
                    for (int p = 0; p < P; ++p) dw[p][j] += __shfl_xor(dw[p][j], o, 64);
                }
            }
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int c0 = (lane % nq) * 8;                     // (= pm.c0 for the lanes below nq)
        for (int turn = 0; turn < 4; ++turn) {
            if (wave == turn && lane < nq) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    lds[c0 + j] += dg[j];
                    lds[C + c0 + j] += db[j];
                    if (DW) {
#pragma unroll
                        for (int p = 0; p < P; ++p) lds[2 * C + (c0 + j) * P + p] += dw[p][j];
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // any other piece count: the rpb row groups of the workgroup (nq threads with distinct channels each) add one after the other
        // -- a fixed order (it was LDS atomics: the order of the hardware, different from run to run)
        for (int turn = 0; turn < pm.rpb; ++turn) {
            if (pm.active && pm.rsub == turn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    lds[pm.c0 + j] += dg[j];
                    lds[C + pm.c0 + j] += db[j];
                    if (DW) {
#pragma unroll
                        for (int p = 0; p < P; ++p) lds[2 * C + (pm.c0 + j) * P + p] += dw[p][j];
                    }
                }
            }
            __syncthreads();
        }
    }
    block_add2_double(s1, s2, sums + 2 * m, dc);  // (contains a __syncthreads: the LDS partials are complete after it)
    float* row = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NV * C);
    for (int i = threadIdx.x; i < NV * C; i += 256) row[i] = lds[i];
    // The PReLU slope's gradient is sum over h <= 0 of dv h with dv = (gamma dy - S1/n - xh S2/n) / sigma, i.e. LINEAR in the three
    // sums taken here; the apply pass used to add it with one fp32 atomic per workgroup on ONE address (1 600 per launch, 28
    // launches: 0.33 ms of the 4.05-ms step went into that queue).  Three floats per workgroup behind the partial rows instead.
    __shared__ float qred[3][4];
    qa = wave_sum(qa); qb = wave_sum(qb); qc = wave_sum(qc);
    if ((threadIdx.x & 63) == 0) { qred[0][threadIdx.x >> 6] = qa; qred[1][threadIdx.x >> 6] = qb; qred[2][threadIdx.x >> 6] = qc; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float* extra = part + (size_t)gridDim.x * gridDim.y * (NV * C);
        extra[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
            qred[threadIdx.x][0] + qred[threadIdx.x][1] + qred[threadIdx.x][2] + qred[threadIdx.x][3];
    }
}

// out[c] += sum over rows of part[row][c]   (grid = (ceil(ncols / 256), row splits); out zeroed by the caller)
__global__ __launch_bounds__(256) void ctn_colsum_kernel(const float* __restrict__ part, int nrows, int ncols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    float s = 0.f;
    for (int r = blockIdx.y; r < nrows; r += gridDim.y) s += part[(size_t)r * ncols + c];
    atomicAdd(&out[c], s);
}

template <int P, bool DW>
__global__ __launch_bounds__(256) void ctn_gln_bwd_apply_kernel(const bf16_raw* __restrict__ g, const bf16_raw* __restrict__ h,
                                                                const float* __restrict__ slope, const double* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ Wd, int dil,
                                                                const double* __restrict__ sums, int K, int C, bf16_raw* __restrict__ dh,
                                                                float* __restrict__ dslope, const float* __restrict__ part, int nrows,
                                                                int ncols, float* __restrict__ gch, int det) {
    // first: the column sums of the reduce kernel's partial rows (gch += sum_r part[r][c]; it was a launch of its own, 28 per step, on
    // the dependent chain), spread over this launch's workgroups as units of (256 columns, one of <= 64 row groups)
    {
        const int nblk = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
        const int ncb = (ncols + 255) >> 8;
        int rg = nrows >> 3;
        rg = rg > 64 ? 64 : (rg < 1 ? 1 : rg);
        if (det) rg = 1;             // deterministic schedule: one unit per 256 columns adds ALL rows in row order, one add per column
        for (int u = bid; u < ncb * rg; u += nblk) {
            const int c = (u % ncb) * 256 + threadIdx.x, r0 = u / ncb;
            if (c < ncols) {
                float acc = 0.f;
                for (int r = r0; r < nrows; r += rg) acc += part[(size_t)r * ncols + c];
                atomicAdd(&gch[c], acc);
            }
        }
    }
    const int m = blockIdx.y, nq = C >> 3;
    const PieceMap pm = piece_map(nq);
    const float a = slope[0];
    float mu, rs;
    gln_moments(stats, m, (long)K * C, mu, rs);
    const double n = (double)K * C;
    const float k1 = (float)(sums[2 * m] / n), k2 = (float)(sums[2 * m + 1] / n);
    const bf16_raw* gb = g + (long)m * K * C + pm.c0;
    const bf16_raw* hb = h + (long)m * K * C + pm.c0;
    bf16_raw* out = dh + (long)m * K * C + pm.c0;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        // slope gradient from the reduce pass's three sums per workgroup (rows of `part` = M utterances x gx workgroups each)
        const int M_ = gridDim.y, gx = nrows / M_;
        const float* extra = part + (size_t)nrows * ncols;
        double acc = 0.0;
        for (int mm = threadIdx.x; mm < M_; mm += 256) {
            double qa = 0.0, qb = 0.0, qc = 0.0;
            for (int i = 0; i < gx; ++i) {
                const float* e = extra + ((size_t)mm * gx + i) * 3;
                qa += e[0]; qb += e[1]; qc += e[2];
            }
            float mu_, rs_;
            gln_moments(stats, mm, (long)K * C, mu_, rs_);
            acc += (double)rs_ * (qa - sums[2 * mm] / n * qb - sums[2 * mm + 1] / n * qc);
        }
        __shared__ double dred[4];
        acc = wave_sum_d(acc);
        if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(dslope, (float)(dred[0] + dred[1] + dred[2] + dred[3]));
    }
    if (pm.active) {
        float gm[8], wd[DW ? P : 1][8];
        ld8f(gamma + pm.c0, gm);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int p = 0; p < (DW ? P : 1); ++p) wd[p][j] = DW ? Wd[(pm.c0 + j) * P + p] : 0.f;
        for (int t = blockIdx.x * pm.rpb + pm.rsub; t < K; t += gridDim.x * pm.rpb) {
            const C8 x = ld8(hb + (long)t * C);
            C8 dy;
            if (!DW) {
                dy = ld8(gb + (long)t * C);
            } else {
                C8 gq[P];
                bool ok[P];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int tt = t - (p - P / 2) * dil;
                    ok[p] = tt >= 0 && tt < K;
                    gq[p] = ld8(gb + (long)(ok[p] ? tt : t) * C);
                }
                dy = zero8();
#pragma unroll
                for (int p = 0; p < P; ++p)
                    if (ok[p]) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) dy.v[j] += wd[p][j] * gq[p].v[j];
                    }
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (prelu(x.v[j], a) - mu) * rs;
                const float dv = (gm[j] * dy.v[j] - k1 - xh * k2) * rs;
                o[j] = x.v[j] > 0.f ? dv : a * dv;
            }
            st8(out + (long)t * C, o);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// decoder: one wave per frame (m, k), both / all speakers in the wave.  V^T in LDS ([n][AL], pitch AL + 1).
//   sw[c][n] = w[n] relu(mlin[c*N + n]);  frame[c][l] = sum_n sw[c][n] V[l][n];  out[m][c][a][k*step + l'] += frame (2 adds per
//   sample: fp32 addition of two terms commutes, so the atomics are deterministic; caller zeroes `out`)
// ------------------------------------------------------------------------------------------------------------------
template <int MC>        // channels per lane: N <= 64 MC (4: N <= 256, 8: N <= 512)
__global__ __launch_bounds__(256) void ctn_decoder_fwd_kernel(const float* __restrict__ w, const bf16_raw* __restrict__ mlin,
                                                              const float* __restrict__ V /*[ac*L][N]*/, int M, int K, int N, int L, int ac,
                                                              int Cs, int T, float* __restrict__ out /*[M][Cs][ac][T]*/) {
    extern __shared__ float smem[];
    const int AL = ac * L;
    float* sV = smem;                                  // [AL][N + 1]
    float* ssw = smem + (size_t)AL * (N + 1);          // per wave [N]
    for (int i = threadIdx.x; i < AL * N; i += 256) { const int l = i / N, n = i - l * N; sV[l * (N + 1) + n] = V[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mysw = ssw + wave * N;
    const int step = L / 2;
    const long frames = (long)M * K;
    for (long fr = (long)blockIdx.x * 4 + wave; fr < frames; fr += (long)gridDim.x * 4) {
        const int m = (int)(fr / K), k = (int)(fr - (long)m * K);
        for (int c = 0; c < Cs; ++c) {
            for (int n = lane; n < N; n += 64) {
                const float ml = bf2f(mlin[fr * ((long)Cs * N) + c * N + n]);
                mysw[n] = w[fr * N + n] * (ml > 0.f ? ml : 0.f);
            }
            // the LDS queue of a wave is in order: no s_barrier needed inside the wave, only a compiler-level barrier so that
            // the reads below are not scheduled above the other lanes' stores
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int al = lane; al < AL; al += 64) {
                float acc = 0.f;
                for (int n = 0; n < N; ++n) acc += mysw[n] * sV[al * (N + 1) + n];
                const int a = al / L, l = al - a * L;
                const long t = (long)k * step + l;
                if (t < T) atomicAdd(&out[(((long)m * Cs + c) * ac + a) * T + t], acc);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// The same on the MFMA pipe (N = 32 KS channels, one audio channel, L <= 16 TA samples per frame, 16 % speakers == 0):
//   D[l][row] = sum_n V[l][n] sw[row][n],   rows = (frame, speaker) pairs, 16 per tile,
// with V as the A operand (held in registers for the whole launch) and sw built in registers from w (fp32) and relu(mlin) (bf16)
// as the B operand -- no serial dot products (the wave-per-frame kernel above spends 128 dependent LDS-read + FMA steps per frame
// and speaker on 40 of its 64 lanes: 162 us at the C4 shape against 60 MB of operands).  Both operands are split into bf16 high
// and low parts (hi hi + lo hi + hi lo: three MFMAs, error ~2^-16): this is the network's output, it stays at fp32 accuracy.
// Overlap-add WITHOUT atomics: a tile holds FT = 16 / speakers consecutive frames of one utterance, first frame 7j - 1 for tile
// j (FT = 8), and emits the FT - 1 hops of L / 2 samples that lie completely inside it (hop b = upper half of frame b - 1 + lower half
// of frame b, through a wave-private LDS image): every output sample is written exactly once with plain, contiguous stores, so
// `out` needs no zeroing.  (With 4.1 M scattered fp32 atomics the kernel took 96 us, with stores in their place 29 us.)
template <int KS, int TA>
__global__ __launch_bounds__(256) void ctn_decoder_fwd_mfma_kernel(const float* __restrict__ w, const bf16_raw* __restrict__ mlin,
                                                                   const float* __restrict__ V /*[L][N]*/, int M, int K, int L, int Cs, int T,
                                                                   float* __restrict__ out /*[M][Cs][T]*/) {
    constexpr int N = 32 * KS, LP = 16 * TA + 1;
    __shared__ float img[4][16][LP];
    const int lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    auto split8 = [](const float (&x)[8], bf16x8& hi, bf16x8& lo) {
        unsigned h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_raw h0 = f2bf(x[2 * i]), h1 = f2bf(x[2 * i + 1]);
            h[i] = (unsigned)h0 | ((unsigned)h1 << 16);
            l[i] = pack_bf2(x[2 * i] - bf2f(h0), x[2 * i + 1] - bf2f(h1));
        }
        hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
        lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
    };
    bf16x8 vhi[TA][KS], vlo[TA][KS];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int al = 16 * ta + c16;
            float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (al < L) {
                const float4 v0 = *reinterpret_cast<const float4*>(V + (size_t)al * N + 32 * ks + 8 * g);
                const float4 v1 = *reinterpret_cast<const float4*>(V + (size_t)al * N + 32 * ks + 8 * g + 4);
                x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
            }
            split8(x, vhi[ta][ks], vlo[ta][ks]);
        }
    const int step = L / 2, FT = 16 / Cs, HT = FT - 1;              // frames per tile, hops emitted per tile
    const int tpu = (K + 1 + HT - 1) / HT;                          // tiles per utterance: hops 0 .. K
    const long ntiles = (long)M * tpu;
    const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
    const int fi = c16 / Cs, c = c16 - fi * Cs;                     // this lane's row: frame inside the tile, speaker
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const int m = (int)(tile / tpu), j = (int)(tile - (long)m * tpu);
        const int k = j * HT - 1 + fi;                              // frame of this lane's row
        const bool rok = k >= 0 && k < K;
        const long fr = (long)m * K + (rok ? k : 0);
        f32x4 acc[TA];
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) acc[ta] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float4 wa[KS][2];
        uint4 ma[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {            // all of the row's loads first
            const float* wp = w + fr * N + 32 * ks + 8 * g;
            wa[ks][0] = *reinterpret_cast<const float4*>(wp);
            wa[ks][1] = *reinterpret_cast<const float4*>(wp + 4);
            ma[ks] = *reinterpret_cast<const uint4*>(mlin + fr * ((long)Cs * N) + (long)c * N + 32 * ks + 8 * g);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned mm[4] = {ma[ks].x, ma[ks].y, ma[ks].z, ma[ks].w};
            const float ww[8] = {wa[ks][0].x, wa[ks][0].y, wa[ks][0].z, wa[ks][0].w, wa[ks][1].x, wa[ks][1].y, wa[ks][1].z, wa[ks][1].w};
            float x[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = __uint_as_float(mm[i] << 16), m1 = __uint_as_float(mm[i] & 0xffff0000u);
                x[2 * i] = rok ? ww[2 * i] * (m0 > 0.f ? m0 : 0.f) : 0.f;
                x[2 * i + 1] = rok ? ww[2 * i + 1] * (m1 > 0.f ? m1 : 0.f) : 0.f;
            }
            bf16x8 shi, slo;
            split8(x, shi, slo);
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) {
                acc[ta] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vhi[ta][ks], shi, acc[ta], 0, 0, 0);
                acc[ta] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vlo[ta][ks], shi, acc[ta], 0, 0, 0);
                acc[ta] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vhi[ta][ks], slo, acc[ta], 0, 0, 0);
            }
        }
        // D row = sample 16 ta + 4 g + q, D column = this lane's row: the tile's frames as an image [row][sample]
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int q = 0; q < 4; ++q) img[wave][c16][16 * ta + 4 * g + q] = acc[ta][q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // hop b = 7 j + bi (bi < HT): samples [b step, (b + 1) step) = frame b - 1 (tile frame bi), upper half + frame b (tile frame bi + 1), lower half
        const int nout = Cs * HT * step;
        for (int o = lane; o < nout; o += 64) {
            const int cc = o / (HT * step), rem = o - cc * (HT * step);
            const int bi = rem / step, sidx = rem - bi * step;
            const int b = j * HT + bi;
            const long t = (long)b * step + sidx;
            if (b <= K && t < T)
                out[((long)m * Cs + cc) * T + t] = img[wave][bi * Cs + cc][step + sidx] + img[wave][(bi + 1) * Cs + cc][sidx];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// gacc: dV [AL][N] (fp32 atomics, caller zeroes).  dmlin [M][K][Cs*N] bf16, dw_dec [M][K][N] fp32 (overwritten)
template <int MC>        // channels per lane: N <= 64 MC (4: N <= 256, 8: N <= 512)
__global__ __launch_bounds__(256) void ctn_decoder_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                              const bf16_raw* __restrict__ mlin, const float* __restrict__ V, int M, int K,
                                                              int N, int L, int ac, int Cs, int T, int frames_per_wave,
                                                              bf16_raw* __restrict__ dmlin, float* __restrict__ dw_dec,
                                                              float* __restrict__ part) {
    extern __shared__ float smem[];
    const int AL = ac * L;
    float* sV = smem;                                  // [AL][N]  (read with n on the lane: conflict free)
    float* sdV = smem + (size_t)AL * N;                // per wave [AL][N] partial dV
    float* sdf = sdV + (size_t)4 * AL * N;             // per wave [AL] dframe
    for (int i = threadIdx.x; i < AL * N; i += 256) sV[i] = V[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mydV = sdV + (size_t)wave * AL * N;
    float* mydf = sdf + wave * AL;
    for (int i = lane; i < AL * N; i += 64) mydV[i] = 0.f;
    __syncthreads();
    const int step = L / 2;
    const long frames = (long)M * K;
    const long f0 = ((long)blockIdx.x * 4 + wave) * frames_per_wave;
    for (long fr = f0; fr < f0 + frames_per_wave && fr < frames; ++fr) {
        const int m = (int)(fr / K), k = (int)(fr - (long)m * K);
        float dwacc[MC];
#pragma unroll
        for (int i = 0; i < MC; ++i) dwacc[i] = 0.f;
        for (int c = 0; c < Cs; ++c) {
            for (int al = lane; al < AL; al += 64) {
                const int a = al / L, l = al - a * L;
                const long t = (long)k * step + l;
                mydf[al] = t < T ? dout[(((long)m * Cs + c) * ac + a) * T + t] : 0.f;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < MC; ++i) {
                const int n = lane + 64 * i;
                if (n >= N) continue;
                const float wv = w[fr * N + n];
                const float ml = bf2f(mlin[fr * ((long)Cs * N) + c * N + n]);
                const float mk = ml > 0.f ? ml : 0.f;
                const float sw = wv * mk;
                float dsw = 0.f;
                for (int al = 0; al < AL; ++al) {
                    const float df = mydf[al];
                    dsw += df * sV[al * N + n];
                    mydV[al * N + n] += df * sw;
                }
                dmlin[fr * ((long)Cs * N) + c * N + n] = f2bf(ml > 0.f ? dsw * wv : 0.f);
                dwacc[i] += dsw * mk;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const int n = lane + 64 * i;
            if (n < N) dw_dec[fr * N + n] = dwacc[i];
        }
    }
    __syncthreads();                                  // one row of partial dV per workgroup, added up by ctn_colsum_kernel
    float* row = part + (size_t)blockIdx.x * (AL * N);
    for (int i = threadIdx.x; i < AL * N; i += 256) row[i] = sdV[i] + sdV[AL * N + i] + sdV[2 * AL * N + i] + sdV[3 * AL * N + i];
}

// Register version of the pass above (AL = ac*L <= 64 at compile time): lane n keeps dV[0..AL)[n] of its NC channels, the
// frame's AL output-gradient samples sit one per lane and are broadcast with v_readlane; only V itself is read from LDS.
template <int AL, int NC>
__global__ __launch_bounds__(256) void ctn_decoder_bwd_reg_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                                  const bf16_raw* __restrict__ mlin, const float* __restrict__ V, int M,
                                                                  int K, int N, int L, int ac, int Cs, int T, int frames_per_wave,
                                                                  bf16_raw* __restrict__ dmlin, float* __restrict__ dw_dec,
                                                                  float* __restrict__ part) {
    extern __shared__ float smem[];
    float* sV = smem;                                  // [AL][N]; reused as the workgroup's dV row at the end
    for (int i = threadIdx.x; i < AL * N; i += 256) sV[i] = V[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int step = L / 2;
    const long frames = (long)M * K;
    const long f0 = ((long)blockIdx.x * 4 + wave) * frames_per_wave;
    float acc[NC][AL];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int al = 0; al < AL; ++al) acc[i][al] = 0.f;
    const int xa = lane / L, xl = lane - xa * L;
    for (long fr = f0; fr < f0 + frames_per_wave && fr < frames; ++fr) {
        const int m = (int)(fr / K), k = (int)(fr - (long)m * K);
        float dwacc[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) dwacc[i] = 0.f;
        for (int c = 0; c < Cs; ++c) {
            const long t = (long)k * step + xl;
            const float dfv = (lane < AL && t < T) ? dout[(((long)m * Cs + c) * ac + xa) * T + t] : 0.f;
            float wv[NC], ml[NC], mk[NC], sw[NC], dsw[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int n = lane + 64 * i;
                wv[i] = n < N ? w[fr * N + n] : 0.f;
                ml[i] = n < N ? bf2f(mlin[fr * ((long)Cs * N) + c * N + n]) : 0.f;
                mk[i] = ml[i] > 0.f ? ml[i] : 0.f;
                sw[i] = wv[i] * mk[i];
                dsw[i] = 0.f;
            }
#pragma unroll
            for (int al = 0; al < AL; ++al) {
                const float df = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dfv), al));
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int n = lane + 64 * i;
                    dsw[i] += df * sV[al * N + (n < N ? n : 0)];
                    acc[i][al] += df * sw[i];
                }
            }
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int n = lane + 64 * i;
                if (n < N) dmlin[fr * ((long)Cs * N) + c * N + n] = f2bf(ml[i] > 0.f ? dsw[i] * wv[i] : 0.f);
                dwacc[i] += dsw[i] * mk[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int n = lane + 64 * i;
            if (n < N) dw_dec[fr * N + n] = dwacc[i];
        }
    }
    __syncthreads();                                   // everybody is done with V
    for (int i = threadIdx.x; i < AL * N; i += 256) sV[i] = 0.f;
    __syncthreads();
    // one wave after the other, plain read-add-write (a wave's lanes touch distinct words): ds_add_f32 from four waves at once on the
    // same words costs ~200 cycles per wave instruction (measured in ctn_encoder_bwd_mfma_kernel)
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int n = lane + 64 * i;
                if (n < N) {
#pragma unroll
                    for (int al = 0; al < AL; ++al) sV[al * N + n] += acc[i][al];
                }
            }
        }
        __syncthreads();
    }
    float* row = part + (size_t)blockIdx.x * (AL * N);
    for (int i = threadIdx.x; i < AL * N; i += 256) row[i] = sV[i];
}

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
static int ctn_check(const char* who, int M, int K, int C) {
    SEHIP_REQUIRE(M > 0 && K > 0, "%s: empty input", who);
    SEHIP_REQUIRE(C >= 8 && C <= 512 && (C & 7) == 0, "%s: channels C=%d must be a multiple of 8 in [8, 512]", who, C);
    return 0;
}
// the backward reductions pay a per-workgroup prologue / epilogue (40 channel constants, the LDS image, a partial row):
// ~512 workgroups in all, each thread then walks 10+ rows
static dim3 ctn_reduce_grid(int M, int K, int C) {
    const int rpb = 256 / (C >> 3) > 0 ? 256 / (C >> 3) : 1;
    static const int total = getenv("SEHIP_CTN_RBLOCKS") ? atoi(getenv("SEHIP_CTN_RBLOCKS")) : 512;     // workgroups (= rows of partials) per launch
    long g = (total + M - 1) / M;
    const long cap = ((long)K + rpb - 1) / rpb;
    if (g > cap) g = cap;
    if (g > 64) g = 64;
    if (g < 1) g = 1;
    return dim3((unsigned)g, (unsigned)M);
}
static dim3 ctn_colsum_grid(int nrows, int ncols) {
    int rs = nrows / 8;                      // >= 8 rows per thread, up to 64 row groups
    if (rs > 64) rs = 64;
    if (rs < 1 || sehip_deterministic()) rs = 1;       // deterministic schedule: one add per column, rows in row order
    return dim3((unsigned)((ncols + 255) / 256), (unsigned)rs);
}
static dim3 ctn_grid(int M, int K, int C) {
    long pieces = (long)K * (C >> 3);
    // 16-byte pieces per thread (C4 step, ms: 4: 3.82, 6: 3.77, 8: 3.78, 16: 3.82, 32: 4.16; 2: 5.6 -- every workgroup pays the
    // per-channel constants and, in the backward apply pass, its share of the column sums)
    static const int rows = getenv("SEHIP_CTN_ROWS") ? atoi(getenv("SEHIP_CTN_ROWS")) : 6;
    long g = (pieces + 256 * rows - 1) / (256 * rows);
    if (g < 1) g = 1;
    if (g > 64) g = 64;
    return dim3((unsigned)g, (unsigned)M);
}

extern "C" int sehip_ctn_encoder_fwd(const float* wav, const float* U, const float* gamma, const float* beta, int M, int ac, int T, int N,
                                     int L, float* w, void* cln_bf16, void* stream) {
    SEHIP_REQUIRE(M > 0 && ac > 0 && T >= L && L >= 2 && (L & 1) == 0, "ctn_encoder_fwd: bad sizes (M=%d ac=%d T=%d L=%d)", M, ac, T, L);
    SEHIP_REQUIRE(N >= 8 && N <= 64 * ENC_MAXC && (N & 7) == 0, "ctn_encoder_fwd: N=%d must be a multiple of 8 up to %d", N, 64 * ENC_MAXC);
    const int K = (T - L) / (L / 2) + 1;
    const size_t lds = (size_t)N * ac * L * sizeof(float);
    SEHIP_REQUIRE(lds <= 64 * 1024, "ctn_encoder_fwd: basis of %zu bytes does not fit the LDS budget", lds);
    static const bool no_mfma = getenv("SEHIP_CTN_NO_MFMA_ENCODER") != nullptr;
    // (the MFMA kernel reads gamma, beta and the rows of U as float4: 16-byte aligned pointers and L % 4 == 0, else the scalar kernel)
    const bool aligned16 = ((((uintptr_t)U) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0 && (L & 3) == 0;
    if (!no_mfma && N == 128 && ac == 1 && L <= 64 && aligned16) {
        static const int cap = getenv("SEHIP_CTN_ENC_WGS") ? atoi(getenv("SEHIP_CTN_ENC_WGS")) : 512;
        long gm = (((long)M * K + 15) / 16 + 3) / 4;
        if (gm > cap) gm = cap;
        ctn_encoder_fwd_mfma_kernel<2><<<(int)gm, 256, 0, (hipStream_t)stream>>>(wav, U, gamma, beta, M, T, K, L, w, (bf16_raw*)cln_bf16);
        SEHIP_CHECK_LAUNCH("ctn_encoder_fwd(mfma)");
        return 0;
    }
    long g = ((long)M * K + 3) / 4;
    if (g > 2048) g = 2048;
    if (N <= 256) ctn_encoder_fwd_kernel<4><<<(int)g, 256, lds, (hipStream_t)stream>>>(wav, U, gamma, beta, M, ac, T, K, N, L, w, (bf16_raw*)cln_bf16);
    else ctn_encoder_fwd_kernel<8><<<(int)g, 256, lds, (hipStream_t)stream>>>(wav, U, gamma, beta, M, ac, T, K, N, L, w, (bf16_raw*)cln_bf16);
    SEHIP_CHECK_LAUNCH("ctn_encoder_fwd");
    return 0;
}

static int ctn_frames_per_wave(long frames) {
    int fpw = (int)((frames + 4 * 512 - 1) / (4 * 512));        // ~512 workgroups
    return fpw < 1 ? 1 : fpw;
}
// floats of scratch the encoder / decoder backward passes need (one row of partial weight-gradient sums per workgroup)
extern "C" long sehip_ctn_codec_bwd_scratch_floats(int M, int K, int N, int L, int ac) {
    const long frames = (long)M * K;
    const int fpw = ctn_frames_per_wave(frames);
    const long grid = (frames + 4L * fpw - 1) / (4L * fpw);
    return grid * ((long)ac * L * N + 2L * N);
}

extern "C" int sehip_ctn_encoder_bwd(const float* wav, const float* w, const void* dcln_bf16, const float* dw_dec, const float* gamma, int M,
                                     int ac, int T, int N, int L, float* gacc, float* scratch, void* stream) {
    SEHIP_REQUIRE(M > 0 && ac > 0 && T >= L && (L & 1) == 0, "ctn_encoder_bwd: bad sizes");
    SEHIP_REQUIRE(N >= 8 && N <= 64 * ENC_MAXC && (N & 7) == 0, "ctn_encoder_bwd: bad N=%d", N);
    SEHIP_REQUIRE(scratch != nullptr, "ctn_encoder_bwd: missing scratch buffer");
    const int K = (T - L) / (L / 2) + 1;
    const size_t lds = ((size_t)4 * N * ac * L + 2 * N) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_encoder_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_encoder_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const long frames = (long)M * K;
    const int fpw = ctn_frames_per_wave(frames);
    const int grid = (int)((frames + 4L * fpw - 1) / (4L * fpw));
    hipStream_t st = (hipStream_t)stream;
    const int ncols = ac * L * N + 2 * N;
    const int AL = ac * L, NC = (N + 63) / 64;
    const size_t rlds = (size_t)ncols * sizeof(float);
    static const bool no_mfma = getenv("SEHIP_CTN_NO_MFMA_ENCODER") != nullptr;
    if (!no_mfma && N == 128 && ac == 1 && L <= 48 && rlds <= 64 * 1024) {
        static const int cap = getenv("SEHIP_CTN_ENCB_WGS") ? atoi(getenv("SEHIP_CTN_ENCB_WGS")) : 512;
        static bool attr_m = false;
        if (!attr_m) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_encoder_bwd_mfma_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_m = true; }
        long tiles4 = ((frames + 31) / 32 + 3) / 4;
        const int gm = (int)(tiles4 < cap ? tiles4 : cap) < grid ? (int)(tiles4 < cap ? tiles4 : cap) : grid;   // rows of `scratch`: at most the old grid
        static const int abl = getenv("SEHIP_CTN_ENCB_ABL") ? atoi(getenv("SEHIP_CTN_ENCB_ABL")) : 0;
        if (abl == 1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_encoder_bwd_mfma_kernel<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); ctn_encoder_bwd_mfma_kernel<3, 1><<<gm, 256, rlds + 4 * (8 * 64 * 16 + 3 * 64 * 32), st>>>(wav, w, (const bf16_raw*)dcln_bf16, dw_dec, gamma, M, T, K, L, scratch); }
        else if (abl == 2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_encoder_bwd_mfma_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); ctn_encoder_bwd_mfma_kernel<3, 2><<<gm, 256, rlds + 4 * (8 * 64 * 16 + 3 * 64 * 32), st>>>(wav, w, (const bf16_raw*)dcln_bf16, dw_dec, gamma, M, T, K, L, scratch); }
        else
        ctn_encoder_bwd_mfma_kernel<3><<<gm, 256, rlds + 4 * (8 * 64 * 16 + 3 * 64 * 32), st>>>(wav, w, (const bf16_raw*)dcln_bf16, dw_dec, gamma, M, T, K, L, scratch);
        ctn_colsum_kernel<<<ctn_colsum_grid(gm, ncols), 256, 0, st>>>(scratch, gm, ncols, gacc);
        SEHIP_CHECK_LAUNCH("ctn_encoder_bwd(mfma)");
        return 0;
    }
#define ENC_REG(AL_, NC_) ctn_encoder_bwd_reg_kernel<AL_, NC_><<<grid, 256, rlds, st>>>(wav, w, (const bf16_raw*)dcln_bf16, dw_dec, gamma, M, ac, T, K, N, L, fpw, scratch)
    if (AL == 40 && NC == 2 && rlds <= 64 * 1024) ENC_REG(40, 2);
    else if (AL == 40 && NC == 1) ENC_REG(40, 1);
    else if (AL == 16 && NC <= 2 && rlds <= 64 * 1024) ENC_REG(16, 2);
    else if (AL == 20 && NC <= 2 && rlds <= 64 * 1024) ENC_REG(20, 2);
    else if (AL == 16 && NC <= 8 && rlds <= 64 * 1024) ENC_REG(16, 8);        // (N = 512, L = 16: the Conv-TasNet paper's encoder)
    else {
        SEHIP_REQUIRE(lds <= 160 * 1024, "ctn_encoder_bwd: %zu bytes of LDS needed", lds);
        if (N <= 256) ctn_encoder_bwd_kernel<4><<<grid, 256, lds, st>>>(wav, w, (const bf16_raw*)dcln_bf16, dw_dec, gamma, M, ac, T, K, N, L, fpw, scratch);
        else ctn_encoder_bwd_kernel<8><<<grid, 256, lds, st>>>(wav, w, (const bf16_raw*)dcln_bf16, dw_dec, gamma, M, ac, T, K, N, L, fpw, scratch);
    }
#undef ENC_REG
    ctn_colsum_kernel<<<ctn_colsum_grid(grid, ncols), 256, 0, st>>>(scratch, grid, ncols, gacc);
    SEHIP_CHECK_LAUNCH("ctn_encoder_bwd");
    return 0;
}

extern "C" int sehip_ctn_gln_stats(const void* h, const float* slope, int M, int K, int C, double* stats /*[M][2], caller zeroes*/, void* stream) {
    if (int e = ctn_check("ctn_gln_stats", M, K, C)) return e;
    const dim3 grid = ctn_grid(M, K, C);
    bool ok;
    const DetCtx dc = sehip_det_ctx((hipStream_t)stream, (size_t)grid.x * grid.y * 2, &ok);
    if (!ok) return -2;
    ctn_gln_stats_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)h, slope, K, C, stats, dc);
    if (int e = sehip_det_finish((hipStream_t)stream, dc, (int)grid.y, (int)grid.x, 2, stats, 2)) return e;
    SEHIP_CHECK_LAUNCH("ctn_gln_stats");
    return 0;
}

extern "C" int sehip_ctn_dwconv_fwd(const void* h1, const float* slope1, const double* stats1, const float* gamma, const float* beta,
                                    const float* Wd, int P, int dilation, const float* slope2, int M, int K, int C, void* h2, double* stats2,
                                    void* stream) {
    if (int e = ctn_check("ctn_dwconv_fwd", M, K, C)) return e;
    SEHIP_REQUIRE(P == 3 || P == 5 || P == 7, "ctn_dwconv_fwd: kernel size P must be 3, 5 or 7 (got %d)", P);
    const dim3 grid = ctn_grid(M, K, C);
    bool ok;
    const DetCtx dc = sehip_det_ctx((hipStream_t)stream, (size_t)grid.x * grid.y * 2, &ok);
    if (!ok) return -2;
#define CTN_DWF(P_) ctn_dwconv_fwd_kernel<P_><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_raw*)h1, slope1, stats1, gamma, beta, Wd, dilation, \
                                                                            slope2, K, C, (bf16_raw*)h2, stats2, dc)
    if (P == 3) CTN_DWF(3);
    else if (P == 5) CTN_DWF(5);
    else CTN_DWF(7);
#undef CTN_DWF
    if (int e = sehip_det_finish((hipStream_t)stream, dc, (int)grid.y, (int)grid.x, 2, stats2, 2)) return e;
    SEHIP_CHECK_LAUNCH("ctn_dwconv_fwd");
    return 0;
}

extern "C" int sehip_ctn_gln_apply(const void* h, const float* slope, const double* stats, const float* gamma, const float* beta, int M, int K,
                                   int C, void* u, void* stream) {
    if (int e = ctn_check("ctn_gln_apply", M, K, C)) return e;
    ctn_gln_apply_kernel<<<ctn_grid(M, K, C), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)h, slope, stats, gamma, beta, K, C, (bf16_raw*)u);
    SEHIP_CHECK_LAUNCH("ctn_gln_apply");
    return 0;
}

// dw != 0: g = dh2 and the incoming gradient is its transposed depthwise convolution (Wd, dilation); gch additionally
// receives dWd.  sums [M][2] and gch / dslope are accumulated with atomics: the caller zeroes them.  scratch: at least
// sehip_ctn_gln_bwd_scratch_floats(M, K, C) floats (the blocks' per-channel partial rows).
extern "C" long sehip_ctn_gln_bwd_scratch_floats(int M, int K, int C) {
    const dim3 grid = ctn_reduce_grid(M, K, C);
    return (long)grid.x * grid.y * (9L * C + 3);          // partial rows of 2 + P <= 9 values per channel + three slope-gradient sums per workgroup
}

extern "C" int sehip_ctn_gln_bwd(const void* g, const void* h, const float* slope, const double* stats, const float* gamma, const float* beta,
                                 const float* Wd, int P, int dilation, int dw, int M, int K, int C, double* sums, float* gch, void* dh,
                                 float* dslope, float* scratch, void* stream) {
    if (int e = ctn_check("ctn_gln_bwd", M, K, C)) return e;
    SEHIP_REQUIRE(!dw || P == 3 || P == 5 || P == 7, "ctn_gln_bwd: kernel size P must be 3, 5 or 7 (got %d)", P);
    SEHIP_REQUIRE((C >> 3) <= 256, "ctn_gln_bwd: too many channels");
    SEHIP_REQUIRE(scratch != nullptr, "ctn_gln_bwd: missing scratch buffer");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = ctn_grid(M, K, C), rgrid = ctn_reduce_grid(M, K, C);
    const int ncols = (2 + (dw ? P : 0)) * C, nrows = (int)(rgrid.x * rgrid.y);
    const size_t lds = (size_t)ncols * sizeof(float);
    bool ok;
    const DetCtx dc = sehip_det_ctx(st, (size_t)rgrid.x * rgrid.y * 2, &ok);
    if (!ok) return -2;
    const int det = sehip_deterministic();
    if (dw) {
#define CTN_GB(P_)                                                                                                                                                         \
    do {                                                                                                                                                                   \
        ctn_gln_bwd_reduce_kernel<P_, true><<<rgrid, 256, lds, st>>>((const bf16_raw*)g, (const bf16_raw*)h, slope, stats, gamma, beta, Wd, dilation, K, C, sums, scratch, dc); \
        if (int e = sehip_det_finish(st, dc, (int)rgrid.y, (int)rgrid.x, 2, sums, 2)) return e;                                                                            \
        ctn_gln_bwd_apply_kernel<P_, true><<<grid, 256, 0, st>>>((const bf16_raw*)g, (const bf16_raw*)h, slope, stats, gamma, Wd, dilation, sums, K, C, (bf16_raw*)dh,      \
                                                                 dslope, scratch, nrows, ncols, gch, det);                                                                  \
    } while (0)
        if (P == 3) CTN_GB(3);
        else if (P == 5) CTN_GB(5);
        else CTN_GB(7);
#undef CTN_GB
    } else {
        ctn_gln_bwd_reduce_kernel<3, false><<<rgrid, 256, lds, st>>>((const bf16_raw*)g, (const bf16_raw*)h, slope, stats, gamma, beta, Wd, dilation, K, C, sums, scratch, dc);
        if (int e = sehip_det_finish(st, dc, (int)rgrid.y, (int)rgrid.x, 2, sums, 2)) return e;
        ctn_gln_bwd_apply_kernel<3, false><<<grid, 256, 0, st>>>((const bf16_raw*)g, (const bf16_raw*)h, slope, stats, gamma, Wd, dilation, sums, K, C, (bf16_raw*)dh, dslope,
                                                                 scratch, nrows, ncols, gch, det);
    }
    SEHIP_CHECK_LAUNCH("ctn_gln_bwd");
    return 0;
}

// mask_nonlinear='softmax' (src/model/conv_tasnet.py:298-299: est_mask = F.softmax(score, dim=1), over the Cs sources): a pass of its own
// between the mask product and the decoder.  The decoder kernels take the result as their "mask logits": their relu is the identity
// on a softmax output, and what sehip_ctn_decoder_bwd returns for it is the gradient of the MASK, which ctn_mask_softmax_bwd turns
// into the gradient of the scores in place:  d score_c = s_c (g_c - sum_c' s_c' g_c').
// rows = M * K frames of [Cs][N] values; one thread per (frame, 8 neighbouring n), the Cs sources in registers.
#define CTN_SOFTMAX_MAX_C 8
__global__ __launch_bounds__(256) void ctn_mask_softmax_fwd_kernel(const bf16_raw* __restrict__ score, long rows, int Cs, int N,
                                                                   bf16_raw* __restrict__ out) {
    const int n8 = N >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * n8; i += (long)gridDim.x * 256) {
        const long r = i / n8;
        const int q = (int)(i - r * n8);
        const bf16_raw* sp = score + r * ((long)Cs * N) + 8 * q;
        C8 x[CTN_SOFTMAX_MAX_C];
        float mx[8], sum[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { mx[j] = -3.0e38f; sum[j] = 0.f; }
#pragma unroll
        for (int c = 0; c < CTN_SOFTMAX_MAX_C; ++c)
            if (c < Cs) {
                x[c] = unpack8(ld8raw(sp + (long)c * N));
#pragma unroll
                for (int j = 0; j < 8; ++j) mx[j] = fmaxf(mx[j], x[c].v[j]);
            }
#pragma unroll
        for (int c = 0; c < CTN_SOFTMAX_MAX_C; ++c)
            if (c < Cs) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { x[c].v[j] = __expf(x[c].v[j] - mx[j]); sum[j] += x[c].v[j]; }
            }
#pragma unroll
        for (int c = 0; c < CTN_SOFTMAX_MAX_C; ++c)
            if (c < Cs) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = x[c].v[j] / sum[j];
                st8(out + r * ((long)Cs * N) + (long)c * N + 8 * q, o);
            }
    }
}

__global__ __launch_bounds__(256) void ctn_mask_softmax_bwd_kernel(const bf16_raw* __restrict__ soft, bf16_raw* __restrict__ g, long rows, int Cs,
                                                                   int N) {
    const int n8 = N >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * n8; i += (long)gridDim.x * 256) {
        const long r = i / n8;
        const int q = (int)(i - r * n8);
        const long base = r * ((long)Cs * N) + 8 * q;
        C8 s[CTN_SOFTMAX_MAX_C], d[CTN_SOFTMAX_MAX_C];
        float dot[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dot[j] = 0.f;
#pragma unroll
        for (int c = 0; c < CTN_SOFTMAX_MAX_C; ++c)
            if (c < Cs) {
                s[c] = unpack8(ld8raw(soft + base + (long)c * N));
                d[c] = unpack8(ld8raw(g + base + (long)c * N));
#pragma unroll
                for (int j = 0; j < 8; ++j) dot[j] += s[c].v[j] * d[c].v[j];
            }
#pragma unroll
        for (int c = 0; c < CTN_SOFTMAX_MAX_C; ++c)
            if (c < Cs) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = s[c].v[j] * (d[c].v[j] - dot[j]);
                st8(g + base + (long)c * N, o);
            }
    }
}

static int ctn_softmax_grid(long items) { long gsz = (items + 255) / 256; return (int)(gsz > 4096 ? 4096 : gsz < 1 ? 1 : gsz); }

extern "C" int sehip_ctn_mask_softmax_fwd(const void* score_bf16, long rows, int Cs, int N, void* out_bf16, void* stream) {
    SEHIP_REQUIRE(rows > 0 && Cs >= 1 && Cs <= CTN_SOFTMAX_MAX_C && N > 0 && N % 8 == 0, "ctn_mask_softmax_fwd: rows=%ld Cs=%d (1..8) N=%d (multiple of 8)",
                  rows, Cs, N);
    ctn_mask_softmax_fwd_kernel<<<ctn_softmax_grid(rows * (N >> 3)), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)score_bf16, rows, Cs, N,
                                                                                                  (bf16_raw*)out_bf16);
    SEHIP_CHECK_LAUNCH("ctn_mask_softmax_fwd");
    return 0;
}

extern "C" int sehip_ctn_mask_softmax_bwd(const void* soft_bf16, void* g_bf16, long rows, int Cs, int N, void* stream) {
    SEHIP_REQUIRE(rows > 0 && Cs >= 1 && Cs <= CTN_SOFTMAX_MAX_C && N > 0 && N % 8 == 0, "ctn_mask_softmax_bwd: rows=%ld Cs=%d (1..8) N=%d (multiple of 8)",
                  rows, Cs, N);
    ctn_mask_softmax_bwd_kernel<<<ctn_softmax_grid(rows * (N >> 3)), 256, 0, (hipStream_t)stream>>>((const bf16_raw*)soft_bf16, (bf16_raw*)g_bf16, rows,
                                                                                                  Cs, N);
    SEHIP_CHECK_LAUNCH("ctn_mask_softmax_bwd");
    return 0;
}

extern "C" int sehip_ctn_decoder_fwd(const float* w, const void* mlin_bf16, const float* V, int M, int K, int N, int L, int ac, int Cs, int T,
                                     float* out /*zeroed by the caller*/, void* stream) {
    SEHIP_REQUIRE(M > 0 && K > 0 && Cs > 0 && ac > 0, "ctn_decoder_fwd: empty input");
    SEHIP_REQUIRE(N >= 8 && N <= 64 * ENC_MAXC, "ctn_decoder_fwd: bad N=%d", N);
    const size_t lds = ((size_t)ac * L * (N + 1) + 4 * N) * sizeof(float);
    SEHIP_REQUIRE(lds <= 64 * 1024, "ctn_decoder_fwd: %zu bytes of LDS needed", lds);
    static const bool no_mfma = getenv("SEHIP_CTN_NO_MFMA_DECODER") != nullptr;
    if (!no_mfma && N == 128 && ac == 1 && L <= 48 && (L & 1) == 0 && Cs >= 1 && Cs <= 8 && 16 % Cs == 0) {
        const int HT = 16 / Cs - 1;
        const long tiles = (long)M * ((K + 1 + HT - 1) / HT);
        // the basis is loaded into registers once per wave: few, long-lived waves
        static const int cap = getenv("SEHIP_CTN_DEC_WGS") ? atoi(getenv("SEHIP_CTN_DEC_WGS")) : 512;
        long gm = (tiles + 3) / 4;
        if (gm > cap) gm = cap;
        ctn_decoder_fwd_mfma_kernel<4, 3><<<(int)gm, 256, 0, (hipStream_t)stream>>>(w, (const bf16_raw*)mlin_bf16, V, M, K, L, Cs, T, out);
        SEHIP_CHECK_LAUNCH("ctn_decoder_fwd(mfma)");
        return 0;
    }
    long g = ((long)M * K + 3) / 4;
    if (g > 2048) g = 2048;
    if (N <= 256) ctn_decoder_fwd_kernel<4><<<(int)g, 256, lds, (hipStream_t)stream>>>(w, (const bf16_raw*)mlin_bf16, V, M, K, N, L, ac, Cs, T, out);
    else ctn_decoder_fwd_kernel<8><<<(int)g, 256, lds, (hipStream_t)stream>>>(w, (const bf16_raw*)mlin_bf16, V, M, K, N, L, ac, Cs, T, out);
    SEHIP_CHECK_LAUNCH("ctn_decoder_fwd");
    return 0;
}

extern "C" int sehip_ctn_decoder_bwd(const float* dout, const float* w, const void* mlin_bf16, const float* V, int M, int K, int N, int L, int ac,
                                     int Cs, int T, void* dmlin_bf16, float* dw_dec, float* gacc, float* scratch, void* stream) {
    SEHIP_REQUIRE(M > 0 && K > 0 && Cs > 0 && ac > 0, "ctn_decoder_bwd: empty input");
    SEHIP_REQUIRE(N >= 8 && N <= 64 * ENC_MAXC, "ctn_decoder_bwd: bad N=%d", N);
    SEHIP_REQUIRE(scratch != nullptr, "ctn_decoder_bwd: missing scratch buffer");
    const size_t lds = ((size_t)5 * ac * L * N + 4 * ac * L) * sizeof(float);
    // (lds: what the wave-per-frame kernel at the end of the chain below needs; the register kernels keep one copy of the basis)
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_decoder_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ctn_decoder_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const long frames = (long)M * K;
    const int fpw = ctn_frames_per_wave(frames);
    const int grid = (int)((frames + 4L * fpw - 1) / (4L * fpw));
    hipStream_t st = (hipStream_t)stream;
    const int ncols = ac * L * N;
    const int AL = ac * L, NC = (N + 63) / 64;
    const size_t rlds = (size_t)ncols * sizeof(float);
#define DEC_REG(AL_, NC_) ctn_decoder_bwd_reg_kernel<AL_, NC_><<<grid, 256, rlds, st>>>(dout, w, (const bf16_raw*)mlin_bf16, V, M, K, N, L, ac, Cs, T, fpw, (bf16_raw*)dmlin_bf16, dw_dec, scratch)
    if (AL == 40 && NC == 2 && rlds <= 64 * 1024) DEC_REG(40, 2);
    else if (AL == 40 && NC == 1) DEC_REG(40, 1);
    else if (AL == 16 && NC <= 2 && rlds <= 64 * 1024) DEC_REG(16, 2);
    else if (AL == 20 && NC <= 2 && rlds <= 64 * 1024) DEC_REG(20, 2);
    else if (AL == 16 && NC <= 8 && rlds <= 64 * 1024) DEC_REG(16, 8);
    else {
        SEHIP_REQUIRE(lds <= 160 * 1024, "ctn_decoder_bwd: %zu bytes of LDS needed", lds);
        if (N <= 256) ctn_decoder_bwd_kernel<4><<<grid, 256, lds, st>>>(dout, w, (const bf16_raw*)mlin_bf16, V, M, K, N, L, ac, Cs, T, fpw,
                                                                       (bf16_raw*)dmlin_bf16, dw_dec, scratch);
        else ctn_decoder_bwd_kernel<8><<<grid, 256, lds, st>>>(dout, w, (const bf16_raw*)mlin_bf16, V, M, K, N, L, ac, Cs, T, fpw,
                                                              (bf16_raw*)dmlin_bf16, dw_dec, scratch);
    }
#undef DEC_REG
    ctn_colsum_kernel<<<ctn_colsum_grid(grid, ncols), 256, 0, st>>>(scratch, grid, ncols, gacc);
    SEHIP_CHECK_LAUNCH("ctn_decoder_bwd");
    return 0;
}
